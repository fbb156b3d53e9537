"""k_edge_res against the round-3 edge kernels in one process (measurement build: HD_EDGE_RES is read at hd_create)."""
import os, sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sizes = [8, 5, 3, 7] if len(sys.argv) < 3 else [30] * int(sys.argv[2])
sd_np = synthetic_state_dict(9, 0, 256, L, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs(sizes, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((len(sizes), 1), 0.5, device=DEV)
outs = {}
for res in ("0", "8", "4"):
    os.environ["HD_EDGE_RES"] = res
    dyn = build_dynamics(sd_np, 256, L); dyn.precision = "fp32"
    with torch.no_grad():
        outs[res] = dyn._forward(t, xh, nm, None, None, None).double().cpu()
for res in ("8", "4"):
    d = outs[res] - outs["0"]
    print("res", res, "L", L, "rel x", float(d[..., :3].norm() / outs["0"][..., :3].norm()), "rel h", float(d[..., 3:].norm() / outs["0"][..., 3:].norm()),
          "max abs", float(d.abs().max()))
d = (outs["8"] - outs["0"])
for c in range(3):
    print("component", c, "rel", float(d[..., c].norm() / outs["0"][..., c].norm()))
print("old", outs["0"][0, :4, :3])
print("new", outs["8"][0, :4, :3])
print("ratio", (outs["8"][0, :4, :3] / outs["0"][0, :4, :3]))
