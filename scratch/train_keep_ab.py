"""Same-box A/B of the training step with the second-layer pre-activations kept (whole-tile forward + loading stage A) against
recomputed (`keep_edge_activations = False`: the column-split / mixed forward where it applies + the recomputing stage A), by batch
size and arithmetic; alternating blocks of 8 steps, two repetitions.  usage: train_keep_ab.py [B ...]"""
import sys, time, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
Bs = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]
N, H, L, DEV = 30, 256, 6, "cuda:0"
m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
m = m.to(DEV).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
for B in Bs:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, 3, generator=g); x = x - x.mean(1, keepdim=True)
    h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
    nm = torch.ones(B, N, 1, dtype=torch.bool); em = ~torch.eye(N, dtype=torch.bool)[None].expand(B, N, N)
    batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.contiguous().to(DEV), "node_feature": h.to(DEV)}
    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.training_step(batch, 0); loss.backward(); opt.step()
    def timed(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    for tp in ("fp32", "bf16x6", "fp16x3"):
        m.dynamics.training_precision = tp
        res = {True: [], False: []}
        for keep in (True, False):
            m.dynamics.keep_edge_activations = keep
            for _ in range(2): step()
        for rep in range(2):
            for keep in (True, False):
                m.dynamics.keep_edge_activations = keep
                res[keep].append(timed(8))
        print(f"B={B:4d} {tp:7s}: kept {min(res[True]):6.2f} ms   recomputed {min(res[False]):6.2f} ms   ({', '.join(f'{a:.2f}/{b:.2f}' for a, b in zip(res[True], res[False]))})", flush=True)
    m.dynamics.keep_edge_activations = True
