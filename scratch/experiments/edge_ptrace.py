"""Cycle stamps of the second tile of every wave of one k_edge_p launch (HD_EDGE_PTRACE=1)."""
import os, sys, ctypes, numpy as np, torch
os.environ["HD_EDGE_PTRACE"] = "1"
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
from hierdiff_amd import _lib
B = 256
sd_np = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs([30] * B, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((B, 1), 0.5, device=DEV)
dyn = build_dynamics(sd_np, 256, 6); dyn.precision = "bf16x3"
topo = dyn.topology(nm, None, B, 30); dyn.sync_weights()
for _ in range(3): o = dyn.forward_with_topology(topo, t, xh, None, None)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(32 * 4096, dtype=np.int64)
n = lib.hd_debug_edge_trace(buf.ctypes.data_as(ctypes.c_void_p), 4096)
tr = buf[: n * 48].reshape(n * 4, 12)
ok = tr[:, 0] > 0
tr = tr[ok]
d = np.diff(tr[:, :11], axis=1)
names = [f"chunk {c}" for c in range(7)] + ["chunk 7", "stage B", "rotate"]
print("waves", len(tr), "tiles per wave", np.bincount(tr[:, 11].astype(int)))
for k, nme in enumerate(names):
    print(f"{nme:8s} mean {d[:, k].mean():8.0f} p10 {np.percentile(d[:, k], 10):8.0f} p50 {np.percentile(d[:, k], 50):8.0f} p90 {np.percentile(d[:, k], 90):8.0f}")
print("tile total mean", (tr[:, 10] - tr[:, 0]).mean())
