"""fp16 two-way split ("fp16x3": hH + hL + lH, fp32 accumulation) next to the bf16 splits and plain fp32: truncation error of
sum_k P[e][k] W[c][k] against fp64.  Variants: l pieces as plain fp16 (subnormals kept / flushed), W scaled by a power of two
so that its largest element sits at 2^14 (exact, undone afterwards), a fourth term lL."""
import numpy as np, torch
torch.manual_seed(0)
E, H = 4096, 256
def run(gain_p, gain_w, tag):
    pre = torch.randn(E, H) * 2.0
    P = torch.nn.functional.silu(pre) * gain_p
    W = (torch.rand(H, H) * 2 - 1) / 16 * gain_w
    ref = P.double() @ W.double().t()
    rel = lambda y: float((y.double() - ref).norm() / ref.norm())
    bf = lambda x: x.to(torch.bfloat16).to(torch.float32)
    def f16(x, flush=False):
        y = x.to(torch.float16)
        if flush:
            y = torch.where(y.abs() < 2.0 ** -14, torch.zeros_like(y), y)
        return y.to(torch.float32)
    def split(x, n, cv):
        parts, r = [], x.clone()
        for _ in range(n):
            p = cv(r); parts.append(p); r = r - p
        return parts
    out = {"fp32 matmul": rel(P @ W.t())}
    for n, terms, name in ((2, [(0, 0), (1, 0), (0, 1)], "bf16x3"), (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)], "bf16x6")):
        p, w = split(P, n, bf), split(W, n, bf)
        out[name] = rel(sum(p[i] @ w[j].t() for i, j in terms))
    for flush in (False, True):
        for scale_w in (False, True):
            sw = 2.0 ** (14 - int(np.floor(np.log2(float(W.abs().max()))))) if scale_w else 1.0
            cv = lambda x: f16(x, flush)
            p, w = split(P, 2, cv), split(W * sw, 2, cv)
            for terms, name in (([(0, 0), (1, 0), (0, 1)], "fp16x3"), ([(0, 0), (1, 0), (0, 1), (1, 1)], "fp16x4")):
                acc = sum(p[i] @ w[j].t() for i, j in terms) / sw
                out[f"{name}{' W@2^14' if scale_w else ''}{' (subnormals flushed)' if flush else ''}"] = rel(acc)
    print(tag)
    for k, v in out.items():
        print(f"   {k:45s} rel-L2 {v:.3e}")
run(1.0, 1.0, "P = SiLU(2 randn), W = U(-1/16, 1/16)")
run(1e-2, 1.0, "P x 1e-2")
run(1.0, 1e-2, "W x 1e-2")
run(30.0, 1.0, "P x 30")
