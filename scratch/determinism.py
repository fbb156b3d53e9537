"""Repeat the full sampler (B=256, N=30, T=200, Philox noise) and compare bitwise: any race in the hand-counted waits
of the edge kernel would show up as run-to-run differences."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
DEV = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = DiffusionQM9(default_config(hidden_nf=256, n_layers=6, timesteps=T))
sd = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.to(DEV)
m.dynamics.precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
print("precision", m.dynamics.precision)
nm = torch.ones(256, 30, 1, dtype=torch.bool, device=DEV)
outs = []
for rep in range(4):
    m.use_graph = bool(rep & 1)
    x, h = m.sample_from_masks(nm, None, None, sample_id_base=0)
    outs.append((x.cpu().numpy(), h.cpu().numpy()))
for k in range(1, 4):
    print("run", k, "vs 0: max |dx|", float(np.abs(outs[k][0] - outs[0][0]).max()), "max |dh|", float(np.abs(outs[k][1] - outs[0][1]).max()))
print("finite", np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][1]).all())
