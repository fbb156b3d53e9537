"""One training step (DiffusionQM9.forward(batch) + backward) at the headline shape; usage: train_step_time.py [B] [L] [fp32|fp16x3]
(third argument: dynamics.training_precision)."""
import sys, time, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N, H, DEV = 30, 256, "cuda:0"
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(DEV).train()
TP = sys.argv[3] if len(sys.argv) > 3 else "fp32"
m.dynamics.training_precision = TP
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N, 3, generator=g); x = x - x.mean(1, keepdim=True)
h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
nm = torch.ones(B, N, 1, dtype=torch.bool); em = ~torch.eye(N, dtype=torch.bool)[None].expand(B, N, N)
batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.contiguous().to(DEV), "node_feature": h.to(DEV)}
from hierdiff_amd.trainer import configure_optimizers
opt, _ = configure_optimizers(m, lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    loss = m.training_step(batch, 0)
    loss.backward()
    opt.step()
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
with torch.no_grad():
    t1 = time.perf_counter(); 
    for _ in range(5): m.forward(batch)
    torch.cuda.synchronize(); dv = (time.perf_counter() - t1) / 5
print(f"B={B} N={N} H={H} L={L} training_precision={TP}: training step (forward + backward + AdamW) {dt*1e3:.1f} ms = {B/dt:.0f} molecules/s; "
      f"validation NLL (2 forwards, no grad) {dv*1e3:.1f} ms; loss {loss.item():.3f}; peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
