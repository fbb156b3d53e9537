"""Device launches of ONE training step by kernel name and by the autograd / module region that issued them (torch profiler).
usage: train_launch_census.py [B] [training_precision]"""
import sys, collections, torch
sys.path.insert(0, '.')
from torch.profiler import ProfilerActivity, profile, record_function
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.trainer import configure_optimizers
from hierdiff_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
TP = sys.argv[2] if len(sys.argv) > 2 else "fp32"
N, H, L, DEV = 30, 256, 6, "cuda:0"
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(DEV).train()
m.dynamics.training_precision = TP
g = torch.Generator().manual_seed(0)
x = torch.randn(B, N, 3, generator=g); x = x - x.mean(1, keepdim=True)
h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
batch = {"positions": x.to(DEV), "atom_mask": torch.ones(B, N, 1, dtype=torch.bool, device=DEV),
         "edge_mask": (~torch.eye(N, dtype=torch.bool))[None].expand(B, N, N).contiguous().to(DEV), "node_feature": h.to(DEV)}
opt, _ = configure_optimizers(m, lr=1e-4)
orig = m.dynamics._forward
def fwd(*a, **k):
    with record_function("REGION dynamics forward"):
        return orig(*a, **k)
m.dynamics._forward = fwd
def step():
    opt.zero_grad(set_to_none=True)
    with record_function("REGION loss forward (incl. dynamics)"):
        loss = m.training_step(batch, 0)
    with record_function("REGION backward"):
        loss.backward()
    with record_function("REGION optimizer"):
        opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
kern = collections.Counter(); ktime = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        name = e.name.split("(")[0][:90]
        kern[name] += 1; ktime[name] += e.device_time if hasattr(e, "device_time") else 0
print(f"B={B} {TP}: {sum(kern.values())} device launches in one step")
for k, c in kern.most_common(40): print(f"  {c:4d}  {k}")
# launches by CPU-side op (the op whose call issued the launch): count hipLaunchKernel children per top-level aten / custom op
ops = collections.Counter()
for e in prof.key_averages():
    if e.key.startswith(("aten::", "_", "Optimizer", "autograd::engine")) and e.count:
        ops[e.key] = e.count
print("ops by count:")
for k, c in ops.most_common(45): print(f"  {c:4d}  {k}")
