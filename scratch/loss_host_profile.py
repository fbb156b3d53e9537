"""Where the host time of one loss evaluation goes (cProfile of DiffusionQM9.forward(batch) under no_grad and of one
training step); usage: loss_host_profile.py [B]."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, H, L, DEV = 30, 256, 6, "cuda:0"
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(DEV).train()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N, 3, generator=g); x = x - x.mean(1, keepdim=True)
h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
nm = torch.ones(B, N, 1, dtype=torch.bool); em = ~torch.eye(N, dtype=torch.bool)[None].expand(B, N, N)
batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.contiguous().to(DEV), "node_feature": h.to(DEV)}
with torch.no_grad():
    for _ in range(2): m.forward(batch)
    torch.cuda.synchronize()
    for mode in ("train", "eval"):
        getattr(m, mode)()
        m.forward(batch); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): m.forward(batch)
        torch.cuda.synchronize()
        print(f"{mode}-mode loss, no grad: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
    m.train()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): m.forward(batch)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True); loss = m.training_step(batch, 0); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter()
torch.cuda.synchronize(); pr.disable()
print(f"training step: host-side {(t1 - t0) / 5 * 1e3:.1f} ms, with GPU drain {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
