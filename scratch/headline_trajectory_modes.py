"""End-of-trajectory agreement of the precision modes at the HEADLINE shape (B = 256, N = 30, H = 256, L = 6, T = 1000, the library's
counter-based noise: the same draws in every mode): final x / h of each opt-in mode against the exact-fp32 run."""
import sys, time, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_diffusion, DEV
from hierdiff_amd.weights import synthetic_state_dict
H, L, T, B, N = 256, 6, 1000, 256, 30
sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 0, 1.0)
nm = torch.ones(B, N, 1, dtype=torch.bool, device=DEV)
out = {}
for prec in ("fp32", "fp16x3", "bf16x6", "bf16x3"):
    model = build_diffusion(sd_np, H, L, T=T, precision=prec)
    model.sample_from_masks(nm[:8].contiguous(), None, None, sample_id_base=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, h = model.sample_from_masks(nm, None, None, sample_id_base=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[prec] = (x.double().cpu(), h.double().cpu())
    print(f"{prec}: {B / dt:.1f} molecules/s; finite {bool(torch.isfinite(x).all() and torch.isfinite(h).all())}; max |x| {float(x.abs().max()):.3g}")
rel = lambda a, b: float((a - b).norm() / b.norm())
for prec in ("fp16x3", "bf16x6", "bf16x3"):
    print(f"{prec} vs fp32 after {T} steps: x rel-L2 {rel(out[prec][0], out['fp32'][0]):.2e}  h rel-L2 {rel(out[prec][1], out['fp32'][1]):.2e}")
