"""Where the HOST spends a training step at the reference's own batch size (conf/dataset/geom_blur.yaml: batch_size 16), where the GPU
is idle most of the time: enqueue time of the sections of a step (no device sync inside), the synced wall time, and the number of
device launches (torch profiler).  usage: train_host_time.py [B] [training_precision]"""
import sys, time, torch
sys.path.insert(0, '.')
from torch.profiler import ProfilerActivity, profile
from hierdiff_amd import DiffusionQM9, default_config
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
FUSED = len(sys.argv) > 3 and sys.argv[3] == "fused"
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True if FUSED else None)
acc = {"dynamics forward": 0.0}
orig = m.dynamics._forward
def timed_fwd(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); acc["dynamics forward"] += time.perf_counter() - t0; return r
m.dynamics._forward = timed_fwd
def step(rec=None):
    t0 = time.perf_counter(); opt.zero_grad(set_to_none=True)
    acc["dynamics forward"] = 0.0
    loss = m.training_step(batch, 0); t1 = time.perf_counter()
    loss.backward(); t2 = time.perf_counter()
    opt.step(); t3 = time.perf_counter()
    if rec is not None:
        rec.append((t1 - t0 - acc["dynamics forward"], acc["dynamics forward"], t2 - t1, t3 - t2))
for _ in range(3): step()
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for _ in range(10): step(rec)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
import numpy as np
r = np.array(rec).mean(0) * 1e3
print(f"B={B} N={N} H={H} L={L} {TP}: step {wall*1e3:.2f} ms synced; host enqueue: loss forward (without the dynamics) {r[0]:.2f} ms, "
      f"dynamics forward {r[1]:.2f}, backward {r[2]:.2f}, optimizer {r[3]:.2f}  (sum {r.sum():.2f})")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
ev = [e for e in prof.key_averages()]
launches = sum(e.count for e in ev if (getattr(e, "device_time_total", 0) or 0) > 0 and e.cpu_time_total == 0)
dev = sum((getattr(e, "device_time_total", 0) or 0) for e in ev if e.cpu_time_total == 0) / 1e3
print(f"device launches per step {launches}, device time {dev:.2f} ms")
rows = sorted(((e.key, e.count, e.self_cpu_time_total) for e in ev if e.self_cpu_time_total > 0), key=lambda t: -t[2])
print("top host-side self times (us):")
for k, c, t in rows[:28]: print(f"  {k[:70]:70s} {c:5d} {t:9.0f}")
