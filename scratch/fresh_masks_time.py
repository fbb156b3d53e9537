"""Training step on never-seen masks: cached twin, device-only masks (mask copy to the host per step), host batches staged one
step ahead (DiffusionQM9.stage_batch) - and the host time of one stage_batch call; usage: fresh_masks_time.py [B] [L] [fp32|bf16x6]."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mode = sys.argv[3] if len(sys.argv) > 3 else "fp32"
H, N = 256, 30
dev = torch.device("cuda:0")
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(dev).train()
m.dynamics.training_precision = mode
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(0)
rng = np.random.Generator(np.random.PCG64(B))
sizes0 = rng.integers(12, N + 1, B)
h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)

def ragged(perm):
    sizes = torch.from_numpy(sizes0[perm])
    nmk = (torch.arange(N)[None, :] < sizes[:, None])
    emk = nmk[:, :, None] & nmk[:, None, :] & ~torch.eye(N, dtype=torch.bool)[None]
    xk = torch.randn(B, N, 3, generator=g) * nmk[..., None]
    xk = xk - (xk.sum(1, keepdim=True) / sizes.view(-1, 1, 1)) * nmk[..., None]
    return {"positions": xk, "atom_mask": nmk[..., None], "edge_mask": emk, "node_feature": h * nmk[..., None]}

on_dev = lambda bt: {k: v.to(dev) for k, v in bt.items()}
K = 12
fresh_host = [ragged(rng.permutation(B)) for _ in range(K)]
fresh = [on_dev(ragged(rng.permutation(B))) for _ in range(K)]
same = on_dev(ragged(np.arange(B)))

import os
_prev = [None]
def step_on(bt):
    if os.environ.get("HD_BOUND") == "1":
        if _prev[0] is not None:
            _prev[0].synchronize()
        _prev[0] = torch.cuda.Event(); _prev[0].record()
    opt.zero_grad(set_to_none=True)
    loss = m.training_step(bt, 0)
    loss.backward()
    opt.step()

def timed(batches):
    for bt in batches[:2]:
        step_on(bt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for bt in batches[2:]:
        step_on(bt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (len(batches) - 2) * 1e3

def timed_staged(batches):
    cur = m.stage_batch(batches[0], dev); host = []
    for k in range(len(batches)):
        if k == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        step_on(cur)
        if k + 1 < len(batches):
            h0 = time.perf_counter(); cur = m.stage_batch(batches[k + 1], dev); host.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (len(batches) - 2) * 1e3, float(np.median(host))

for _ in range(6):
    step_on(same)
rc, rf, rs, hs = [], [], [], 0.0
for rep in range(3):                                   # interleaved repeats, best of three (the first second of a process is noisy)
    rc.append(timed([same] * K))
    rf.append(timed(fresh))
    a, hs = timed_staged(fresh_host)
    rs.append(a)
    fresh = [on_dev(ragged(rng.permutation(B))) for _ in range(K)]
    fresh_host = [ragged(rng.permutation(B)) for _ in range(K)]
f = lambda v: "/".join(f"{x:.2f}" for x in v)
print(f"B={B} L={L} {mode}: ms per step, best of 3 [all]: cached masks {min(rc):.2f} [{f(rc)}]; device-only fresh masks {min(rf):.2f} [{f(rf)}]; "
      f"staged host batches {min(rs):.2f} [{f(rs)}] (host time of one stage_batch {hs:.2f} ms)")
