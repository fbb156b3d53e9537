"""Per-step wall times of the staged fresh-mask loop (40 steps) with the pieces of stage_batch timed on the host."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
import hierdiff_amd.dynamics as D
B, L, H, N = 256, 6, 256, 30
dev = torch.device("cuda:0")
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(dev).train()
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
K = 40
batches = [ragged(rng.permutation(B)) for _ in range(K)]
pieces = {"topo": [], "copy": []}
orig_tbc = D.EGNN_dynamics_QM9._topology_by_content
def tbc(self, *a):
    t0 = time.perf_counter(); r = orig_tbc(self, *a); pieces["topo"].append((time.perf_counter() - t0) * 1e3); return r
D.EGNN_dynamics_QM9._topology_by_content = tbc
def step_on(bt):
    opt.zero_grad(set_to_none=True)
    loss = m.training_step(bt, 0)
    loss.backward()
    opt.step()
cur = m.stage_batch(batches[0], dev)
walls, hosts, launches = [], [], []
torch.cuda.synchronize()
for k in range(K):
    t0 = time.perf_counter()
    step_on(cur)
    t1 = time.perf_counter()
    if k + 1 < K:
        cur = m.stage_batch(batches[k + 1], dev)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    walls.append((t3 - t0) * 1e3); launches.append((t1 - t0) * 1e3); hosts.append((t2 - t1) * 1e3)
f = lambda v: " ".join(f"{x:.1f}" for x in v)
print("step wall ms :", f(walls))
print("launch host ms:", f(launches))
print("stage host ms :", f(hosts))
print("topology ms   :", f(pieces["topo"]))
import cProfile, pstats
batches = [ragged(rng.permutation(B)) for _ in range(20)]
cur = m.stage_batch(batches[0], dev)
pr = cProfile.Profile(); pr.enable()
for k in range(20):
    step_on(cur)
    if k + 1 < 20:
        cur = m.stage_batch(batches[k + 1], dev)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
