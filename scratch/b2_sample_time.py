"""Wall time of a full T=1000 sample at small batches (graph replay), per precision: usage b2_sample_time.py [B ...]"""
import sys, time, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_diffusion, DEV
from hierdiff_amd.weights import synthetic_state_dict
Bs = [int(v) for v in sys.argv[1:]] or [2]
sd = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
for prec in ("fp32", "fp16x3", "bf16x3"):
    m = build_diffusion(sd, 256, 6, T=1000, precision=prec)
    for B in Bs:
        nm = torch.ones(B, 30, 1, dtype=torch.bool, device=DEV)
        m.sample_from_masks(nm, None, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, h = m.sample_from_masks(nm, None, None)
        x.cpu(); dt = time.perf_counter() - t0
        print(f"{prec} B={B}: {dt:.4f} s per 1000-step batch = {dt / 1001 * 1e3:.4f} ms per forward+step, {B / dt:.2f} molecules/s")
