"""Cost of the content check (round 6): per-call `_forward` at small and headline batch with the guard on / off (off = ImageGuard.valid
patched to the key-only compare), and the bare digest call."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hierdiff_amd import _lib, DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict

dev = "cuda:0"
m = DiffusionQM9(default_config())
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, 256, 6, 2, True, 1, 0.02).items()})
m = m.to(dev).eval()
params = list(m.dynamics.egnn.parameters())
for _ in range(3):
    _lib.params_digest(params)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    _lib.params_digest(params)
print(f"params_digest over {sum(p.numel() for p in params) * 4 / 1e6:.1f} MB, idle stream: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
key_only = lambda self, key, tensors: self.key is not None and key == self.key
real = _lib.ImageGuard.valid
for prec in ("fp32", "fp16x3"):
    m.dynamics.precision = prec
    for B in (2, 16, 256):
        nm = torch.ones(B, 30, 1, dtype=torch.bool, device=dev)
        xh = torch.randn(B, 30, 11, device=dev)
        t = torch.full((B, 1), 0.5, device=dev)
        res = {}
        for name, fn in (("guard", real), ("key-only", key_only), ("guard2", real), ("key-only2", key_only)):
            _lib.ImageGuard.valid = fn
            with torch.no_grad():
                for _ in range(5):
                    m.dynamics._forward(t, xh, nm, None, None)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 200 if B < 256 else 40
                for _ in range(n):
                    m.dynamics._forward(t, xh, nm, None, None)
                torch.cuda.synchronize()
                res[name] = (time.perf_counter() - t0) / n * 1e3
        _lib.ImageGuard.valid = real
        print(f"{prec} B={B}: per-call _forward ms " + "  ".join(f"{k} {v:.3f}" for k, v in res.items()))
