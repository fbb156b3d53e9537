"""Per-kernel-family time of one forward via the library's HIP-event profiler."""
import sys, os, ctypes as C, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd import _lib
from hierdiff_amd.weights import synthetic_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sd_np = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs([30] * B, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((B, 1), 0.5, device=DEV)
dyn = build_dynamics(sd_np, 256, 6); dyn.precision = prec
topo = dyn.topology(nm, None, B, 30); dyn.sync_weights()
lib = _lib.load(); h = dyn._handle()
for _ in range(3): dyn.forward_with_topology(topo, t, xh, None, None)
torch.cuda.synchronize()
lib.hd_profile_enable(h, 7)
for _ in range(10): dyn.forward_with_topology(topo, t, xh, None, None)
ms = (C.c_double * 3)(); cnt = (C.c_longlong * 3)()
lib.hd_profile_read(h, ms, cnt)
print(prec, "ABL", os.environ.get("HD_ABLATE"), " edge %.1f us/launch (%d)  gemm %.1f us/launch (%d)  other %.1f us/launch (%d)" % (
    ms[0] / cnt[0] * 1e3, cnt[0], ms[1] / cnt[1] * 1e3, cnt[1], ms[2] / cnt[2] * 1e3, cnt[2]))
