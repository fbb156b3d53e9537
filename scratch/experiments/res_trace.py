"""Per-wave cycle stamps of one steady-state tile of the last k_edge_res launch of a forward (measurement build, HIERDIFF_LIB):
where does a tile's time go?  usage: python scratch/res_trace.py [B]"""
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
lib.hd_debug_node_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
NWV = 8 if os.environ.get("HD_EDGE_RES") == "8" else 4
tr = buf.reshape(512, 8, NS)[:256, :NWV].astype(np.float64)
names = ["0 top of tile", "1 first A request", "2 MFMAs done", "3 epi1 done (partial dots out)", "4 next operand built", "5 barrier + row dot",
         "6 epi2 done"]
ok = (tr[..., :7] > 0).all(axis=(1, 2))
tr = tr[ok]
print("B", B, "workgroups with a third tile:", int(ok.sum()))
for k in range(1, 7):
    d = tr[..., k] - tr[..., k - 1]
    print(f"{names[k-1]:32s} -> {names[k]:32s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p50 {np.percentile(d, 50):8.0f}  p90 {np.percentile(d, 90):8.0f}")
d = tr[..., 6] - tr[..., 0]
print("tile total (0 -> 6): mean %.0f p50 %.0f p90 %.0f" % (d.mean(), np.percentile(d, 50), np.percentile(d, 90)))
w = tr[..., 4]
print("arrival skew at the barrier over a workgroup's 8 waves (max - min of stamp 4): mean %.0f p90 %.0f" % ((w.max(1) - w.min(1)).mean(), np.percentile(w.max(1) - w.min(1), 90)))
for blk in (0, 1):
    base = tr[blk][:, :7].min()
    print("workgroup", blk, "stamps relative to its first (rows = waves, columns = stamps 0..6):")
    for w in range(tr.shape[1]):
        print("  wave", w, " ".join(f"{int(v - base):7d}" for v in tr[blk][w, :7]))
