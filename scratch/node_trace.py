"""Per-wave cycle stamps of the last k_node_f32 launch of one forward: where does a 32-row tile's time go?
Needs the measurement build (python -m hierdiff_amd.build --debug-kernels) selected with HIERDIFF_LIB."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
from hierdiff_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sd_np = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs([30] * B, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((B, 1), 0.5, device=DEV)
dyn = build_dynamics(sd_np, 256, 6); dyn.precision = "fp32"
topo = dyn.topology(nm, None, B, 30); dyn.sync_weights()
for _ in range(5): o = dyn.forward_with_topology(topo, t, xh, None, None)
torch.cuda.synchronize()
lib = _lib.load()
NS = 12
buf = np.zeros(512 * 8 * NS, dtype=np.int64)
lib.hd_debug_node_trace.restype = ctypes.c_int
n = lib.hd_debug_node_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
nwg = (B * 30 + 31) // 32
tr = buf.reshape(512, 8, NS)[:nwg].astype(np.float64)
t0 = tr[..., 0].min()
print("workgroups", nwg, " launch span (first stamp 0 -> last stamp 11):", tr[..., 11].max() - t0, "ticks")
names = ["0 entry", "1 X stored", "2 barrier (X complete)", "3 M1 done", "4 T stored + barrier", "5 M2 done", "6 h' stored, h_out issued",
         "7 M3(q0) done", "8 AB0 staged+stored", "9 M3(q1) done", "10 AB1 staged+stored", "11 end"]
prev = 0
for k in range(1, NS):
    if not (tr[..., k] > 0).all(): continue
    d = tr[..., k] - tr[..., prev]
    print(f"{names[prev]:28s} -> {names[k]:28s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p50 {np.percentile(d, 50):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    prev = k
print("entry skew over workgroups (stamp 0 - min): p50 %.0f  p90 %.0f  max %.0f" % tuple(np.percentile(tr[..., 0] - t0, [50, 90, 100])))
print("wave total: mean %.0f  max %.0f" % ((tr[..., 11] - tr[..., 0]).mean(), (tr[..., 11] - tr[..., 0]).max()))
