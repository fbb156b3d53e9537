"""hd_gemm_f32 (csrc/k_tgemm.hpp) against torch.matmul (the BLAS library) on the training path's shapes; us per call."""
import sys, time, torch
sys.path.insert(0, '.')
from hierdiff_amd import training as tr
DEV = "cuda:0"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 7680
for (M, N, K) in [(rows, 256, 256), (rows, 512, 256), (rows, 256, 512), (rows, 1024, 256)]:
    X = torch.randn(M, K, device=DEV); W = torch.randn(N, K, device=DEV); b = torch.randn(N, device=DEV); gy = torch.randn(M, N, device=DEV)
    fl = 2.0 * M * N * K
    a = timeit(lambda: tr._linear_fwd(X, W, b)); b_ = timeit(lambda: torch.addmm(b, X, W.t()))
    c = timeit(lambda: tr._linear_dx(gy, W)); d = timeit(lambda: gy @ W)
    e = timeit(lambda: tr._linear_dw(gy, X, True)); f = timeit(lambda: (gy.t() @ X, gy.sum(0)))
    print(f"M={M} N={N} K={K}: fwd own {a:7.1f} us ({fl/a/1e6:6.1f} TF/s) blas {b_:7.1f} | dX own {c:7.1f} blas {d:7.1f} | dW+db own {e:7.1f} (split {tr._split_for(N, K, M)}) blas {f:7.1f}")
R = 223232
G2 = torch.randn(R, 256, device=DEV); P = torch.randn(R, 256, device=DEV)
e = timeit(lambda: tr._linear_dw(G2, P, False), 20)
f = timeit(lambda: torch.bmm(G2.view(32, -1, 256).transpose(1, 2), P.view(32, -1, 256)).sum(0), 20)
print(f"dW2 rows={R}: own {e:7.1f} us ({2.0*R*256*256/e/1e6:6.1f} TF/s, split {tr._split_for(256, 256, R)}) blas split-32 bmm {f:7.1f}")
