"""Where one Edge_denoise.sample_AR step spends its time (host profile + kernel count): beam of 24 half-grown 12-node trees."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from hierdiff_amd.edge_denoise import Edge_denoise, synthetic_edge_denoise_state_dict
dev = torch.device("cuda:0")
H, bs, n = 256, 24, 12
kw = dict(vocab_size=781, in_node_nf=8, hidden_nf=H, out_node_nf=780, context_nf=0)
ed = Edge_denoise(array_dict=None, full_softmax=True, **kw)
ed.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_edge_denoise_state_dict(3, **kw).items()})
ed = ed.to(dev)
rng = np.random.Generator(np.random.PCG64(5))
adj = torch.zeros(bs, n, n)
for b in range(bs):
    for v in range(1, 6):
        p = int(rng.integers(0, v)); adj[b, v, p] = adj[b, p, v] = 1
feat = torch.from_numpy(rng.standard_normal((bs, n, 10)).astype(np.float32))
feat[:, :, 9] = torch.from_numpy(rng.integers(0, 780, (bs, n)).astype(np.float32))
beam = {'node_feat': [feat.to(dev), torch.ones(bs, n, 10, device=dev)], 'node_pos': (torch.randn(bs, n, 3) * 1.5).to(dev),
        'search_adj_matrix': adj.to(dev), 'edge_mask': (1 - torch.eye(n))[None].expand(bs, n, n).contiguous().to(dev)}
for _ in range(3): ed.sample_AR(beam)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ed.sample_AR(beam)
torch.cuda.synchronize(); print(f"sample_AR: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): ed.sample_AR(beam)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
