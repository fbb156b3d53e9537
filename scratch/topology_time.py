"""Host cost of a fresh topology (new masks every training batch): hd_topology_create incl. the mask D->H copies, the host
layout and the device allocations / uploads; usage: topology_time.py [B] [N]."""
import sys, time, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from hierdiff_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dyn = build_dynamics(synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0), 256, 6)
dyn.sync_weights()
g = torch.Generator().manual_seed(0)
for canonical in (True, False):
    ts = []
    for k in range(6):
        n = torch.randint(N // 2, N + 1, (B,), generator=g)
        nm = (torch.arange(N)[None, :] < n[:, None]).unsqueeze(-1).to(DEV)
        em = None if canonical else (nm[:, :, 0][:, :, None] & nm[:, :, 0][:, None, :] & ~torch.eye(N, dtype=torch.bool, device=DEV)[None])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        topo = dyn.topology(nm, em, B, N)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"B={B} N={N} {'canonical' if canonical else 'explicit'} edge mask: fresh topology {min(ts[1:]):.2f} ms (min of 5), first {ts[0]:.2f} ms; {topo.info()}")
