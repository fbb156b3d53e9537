"""Truncation error of multi-term bf16 splits of an fp32 contraction (no GPU needed): how far is
sum_k P[e][k] W[c][k] computed from bf16 pieces from the exact (fp64) value, next to plain fp32 accumulation?
x3 = hh + lh + hl (2-way split), x6 = hh + hm + mh + mm + hl + lh (3-way split).  Accumulation of the kept products is done
in fp64 here (the MFMA accumulates in fp32 in every mode, which adds the same ~1e-7 to all of them), so the numbers isolate
what the dropped terms cost."""
import numpy as np, torch
torch.manual_seed(0)
E, H = 4096, 256
pre = torch.randn(E, H) * 2.0
P = torch.nn.functional.silu(pre)
W = (torch.rand(H, H) * 2 - 1) / 16
def bf(x): return x.to(torch.bfloat16).to(torch.float32)
def split(x, n):
    parts, r = [], x.clone()
    for _ in range(n):
        p = bf(r); parts.append(p); r = r - p
    return parts
ref = P.double() @ W.double().t()
def rel(y): return float((y.double() - ref).norm() / ref.norm()), float((y.double() - ref).abs().max() / ref.abs().max())
print("fp32 matmul (CPU BLAS)      rel-L2 %.3e  max/max %.3e" % rel(P @ W.t()))
for n, terms, name in ((2, [(0, 0), (1, 0), (0, 1)], "bf16x3"), (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)], "bf16x6"),
                       (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0), (1, 2), (2, 1), (2, 2)], "bf16x9")):
    p, w = split(P, n), split(W, n)
    acc = sum(p[i].double() @ w[j].double().t() for i, j in terms)
    print("%s split, exact accumulation rel-L2 %.3e  max/max %.3e   -> rounded to fp32: %.3e" % ((name,) + rel(acc) + (rel(acc.float())[0],)))
