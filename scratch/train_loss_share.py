"""Where the small launches of a training step are: the full step, the step with the variational loss replaced by a plain
sum over the network output (dynamics + optimizer only), and the loss evaluated alone on a detached network output.
usage: train_loss_share.py [B]"""
import sys, time, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, N, H, DEV = 6, 30, 256, "cuda:0"
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(DEV).train()
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N, 3, generator=g); x = x - x.mean(1, keepdim=True)
h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
nm = torch.ones(B, N, 1, dtype=torch.bool); em = ~torch.eye(N, dtype=torch.bool)[None].expand(B, N, N)
batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.contiguous().to(DEV), "node_feature": h.to(DEV)}
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def full():
    opt.zero_grad(set_to_none=True); loss = m.training_step(batch, 0); loss.backward(); opt.step()
xh = torch.cat([batch["positions"], batch["node_feature"]], 2)
t = torch.full((B, 1), 0.5, device=DEV)
def dyn_only():
    opt.zero_grad(set_to_none=True)
    out = m.dynamics._forward(t, xh, batch["atom_mask"], batch["edge_mask"].reshape(B, N * N), None, None)
    out.sum().backward(); opt.step()
def dyn_fwd_bwd_no_opt():
    for p in m.parameters(): p.grad = None
    out = m.dynamics._forward(t, xh, batch["atom_mask"], batch["edge_mask"].reshape(B, N * N), None, None)
    out.sum().backward()
print(f"B={B}: full step {timed(full):.2f} ms; dynamics fwd+bwd+AdamW with a plain sum as loss {timed(dyn_only):.2f} ms; the same without the optimizer {timed(dyn_fwd_bwd_no_opt):.2f} ms")
from torch.profiler import profile, ProfilerActivity
for name, fn in (("full", full), ("dyn_only", dyn_only)):
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn(); torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type.name == "CUDA"] if hasattr(prof.events()[0], "device_type") else []
    n = len(ev); tot = sum(e.device_time if hasattr(e, "device_time") else e.cuda_time for e in ev)
    small = [e for e in ev if (e.device_time if hasattr(e, "device_time") else e.cuda_time) < 20]
    print(f"{name}: {n} device events, {tot/1e3:.2f} ms device time; {len(small)} of them under 20 us, together {sum((e.device_time if hasattr(e, 'device_time') else e.cuda_time) for e in small)/1e3:.2f} ms")
