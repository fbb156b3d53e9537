import sys, numpy as np, torch
sys.path.insert(0, '.')
from hierdiff_amd import EnVariationalDiffusion, TwoStreamSampler, default_config
from hierdiff_amd.weights import synthetic_state_dict
DEV = "cuda:0"
H, L, T, N = 128, 2, 12, 12
sd = synthetic_state_dict(9, 1, H, L, 2, True, 71, 1.0)
m = EnVariationalDiffusion(default_config(hidden_nf=H, n_layers=L, context_node_nf=1, timesteps=T))
m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd.items()})
m = m.to(DEV).eval()
m.dynamics.precision = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
sizes = torch.tensor([12, 5, 9, 1, 12, 7, 3])
nm = (torch.arange(N)[None, :] < sizes[:, None]).unsqueeze(-1).to(DEV)
ctx = torch.linspace(-0.4, 4.9, 7).view(7, 1, 1).expand(7, N, 1).contiguous().to(DEV)
two = TwoStreamSampler(m)
eq = lambda a, b: (bool(torch.equal(a, b)), float((a - b).abs().max()))
with torch.no_grad():
    x0, _ = m.sample_from_masks(nm, None, ctx, sample_id_base=40)
    nmA, cxA = nm[:4].contiguous(), ctx[:4].contiguous()
    if len(sys.argv) > 2 and sys.argv[2] == "pre":
        xa0, _ = m.sample_from_masks(nmA, None, cxA, sample_id_base=40); print("half-0 alone BEFORE two-stream == full[:4]", eq(xa0, x0[:4]))
    x, _ = two.sample_from_masks(nm, None, ctx, sample_id_base=40); torch.cuda.synchronize()
    print("two == single", eq(x, x0))
    x1, _ = m.sample_from_masks(nm, None, ctx, sample_id_base=40); print("full again == full", eq(x1, x0))
    xa, _ = m.sample_from_masks(nmA, None, cxA, sample_id_base=40); print("half-0 alone AFTER == full[:4]", eq(xa, x0[:4]))
    xa2, _ = m.sample_from_masks(nmA, None, cxA, sample_id_base=40); print("half-0 alone again == previous", eq(xa2, xa))
    m.use_graph = False
    xa3, _ = m.sample_from_masks(nmA, None, cxA, sample_id_base=40); print("half-0 alone, plain launches == full[:4]", eq(xa3, x0[:4]))
    m.use_graph = True
    nmB, cxB = nm[4:].contiguous(), ctx[4:].contiguous()
    xb, _ = m.sample_from_masks(nmB, None, cxB, sample_id_base=44); print("half-1 alone on m == full[4:]", eq(xb, x0[4:]))
    xt, _ = two._twin.sample_from_masks(nmB, None, cxB, sample_id_base=44); print("half-1 on twin again == full[4:]", eq(xt, x0[4:]))
