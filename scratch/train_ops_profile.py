"""aten-level operator counts / device time of one training step (B=256): where the small torch launches come from."""
import sys, torch
sys.path.insert(0, '.')
from torch.profiler import ProfilerActivity, profile
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
B, N, H, L, DEV = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 30, 256, 6, "cuda:0"
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(DEV).train()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N, 3, generator=g); x = x - x.mean(1, keepdim=True)
h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
batch = {"positions": x.to(DEV), "atom_mask": torch.ones(B, N, 1, dtype=torch.bool, device=DEV),
         "edge_mask": (~torch.eye(N, dtype=torch.bool))[None].expand(B, N, N).contiguous().to(DEV), "node_feature": h.to(DEV)}
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    loss = m.training_step(batch, 0); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0), e.cpu_time_total) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[2])
print(f"{'op':60s} {'count':>6s} {'device us':>10s} {'cpu us':>10s}")
for k, c, d, cpu in rows[:45]:
    print(f"{k[:60]:60s} {c:6d} {d:10.0f} {cpu:10.0f}")
# who calls the expensive element-wise / reduction ops: group by (op, innermost repo frame)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
    step(); torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof2.events():
    if e.name in ("aten::sum", "aten::mul", "aten::add_", "aten::copy_", "aten::slice_backward", "aten::fill_", "aten::add", "aten::cat", "aten::zeros", "aten::index", "aten::index_copy", "aten::where"):
        d = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
        fr = [s for s in (e.stack or []) if "/root/repo" in s or "hierdiff_amd" in s]
        key = (e.name, fr[0].split("/")[-1][:70] if fr else "(autograd engine / optimizer)")
        agg[key][0] += 1; agg[key][1] += d
print("---- by call site")
for (n, site), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:22s} {site:72s} {c:5d} {d:9.0f} us")
