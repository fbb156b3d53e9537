"""Per-kernel averages of one stage-2 gcl_full layer forward (bs = 24 graphs of 12 nodes, H = 256): run under
rocprofv3 --kernel-trace --stats."""
import sys, time, torch
sys.path.insert(0, '.')
from hierdiff_amd.stage2 import E_GCL, synthetic_egcl_state_dict
dev = torch.device("cuda:0")
H, bs, n = 256, 24, 12
ar = torch.arange(n)
row = (ar.repeat_interleave(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n)).to(dev)
col = (ar.repeat(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n)).to(dev)
g = torch.Generator().manual_seed(1)
hh = torch.randn(bs * n, H, generator=g).to(dev); xx = torch.randn(bs * n, 3, generator=g).to(dev)
ea = torch.randn(row.numel(), H, generator=g).to(dev)
nmask = torch.ones(bs * n, 1, device=dev); emask = (row != col).float().unsqueeze(1)
lay = E_GCL(H, H, H, edges_in_d=H, attention=True, tanh=True, coords_range=30, edge_update=True)
lay.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_egcl_state_dict(H, H, 0, True, True, 40, coord_gain=0.3).items()})
lay = lay.to(dev)
for _ in range(3): lay(hh, [row, col], xx, edge_attr=ea, node_mask=nmask, edge_mask=emask)
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 50
for _ in range(R): lay(hh, [row, col], xx, edge_attr=ea, node_mask=nmask, edge_mask=emask)
torch.cuda.synchronize(); print(f"layer forward {(time.perf_counter() - t0) / R * 1e3:.3f} ms")
