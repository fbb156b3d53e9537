"""A/B of the node-level GEMMs of a large training batch: k_tgemm (64 x 128 LDS-tiled) vs k_tgemm_rows (row-resident, round 6).
Run with the measurement build: HIERDIFF_LIB=hierdiff_amd/lib/libhierdiff_hip_dbg.so HD_TGEMM_ROWS_MIN=<rows> python scratch/gemm_rows_ab.py
(HD_TGEMM_ROWS_MIN=100000000 = the old kernel everywhere).  Prints us per call for the three Linears of a node update, forward and dX,
and the training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierdiff_amd import training as tr
dev = "cuda:0"
tag = os.environ.get("HD_TGEMM_ROWS_MIN", "default")
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for M in (7680, 3840):
    for N, K in ((512, 256), (256, 512), (256, 256)):
        X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); gy = torch.randn(M, N, device=dev)
        tf = timeit(lambda: tr._linear_fwd(X, W, b))
        ts = timeit(lambda: tr._linear_fwd(X, W, b, tr._EPI_BIAS_SILU2))
        td = timeit(lambda: tr._linear_dx(gy, W))
        fl = 2.0 * M * N * K
        print(f"[rows_min {tag}] M={M} N={N} K={K}: fwd {tf:6.1f} us ({fl / tf / 1e6:5.1f} TFLOP/s)  fwd+silu2 {ts:6.1f}  dX {td:6.1f} us ({fl / td / 1e6:5.1f} TFLOP/s)")
os.system(f"{sys.executable} scratch/train_step_time.py 256 6 fp32 | tail -1")
os.system(f"{sys.executable} scratch/train_step_time.py 256 6 fp16x3 | tail -1")
os.system(f"{sys.executable} scratch/train_step_time.py 128 6 fp32 | tail -1")
