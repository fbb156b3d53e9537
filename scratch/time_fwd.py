import sys, time, os, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sd_np = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs([30] * B, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((B, 1), 0.5, device=DEV)
dyn = build_dynamics(sd_np, 256, 6); dyn.precision = prec
topo = dyn.topology(nm, None, B, 30); dyn.sync_weights()
for _ in range(3): o = dyn.forward_with_topology(topo, t, xh, None, None)
torch.cuda.synchronize(); t0 = time.perf_counter()
R = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for _ in range(R): o = dyn.forward_with_topology(topo, t, xh, None, None)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
print(prec, "B", B, "stagger", os.environ.get("HD_STAGGER"), f"{dt*1e3:.3f} ms/forward")
