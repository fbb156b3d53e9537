import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import egnn_oracle as orc
from hierdiff_amd import EGNN_dynamics_QM9, DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
DEV = "cuda:0"
def rel(a, b): return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
sd_np = synthetic_state_dict(9, 0, 64, 2, 2, True, 1, 1.0)
cfg = orc.DynCfg(in_node_nf=9, hidden_nf=64, n_layers=2)
dyn = EGNN_dynamics_QM9(9, 0, 3, hidden_nf=64, n_layers=2, attention=True, tanh=True, normalization_factor=10)
dyn.load_numpy_state_dict(sd_np, prefix="dynamics."); dyn = dyn.to(DEV)
for n_list, pad in (([1], None), ([1, 1, 1], None), ([1], 5), ([2], None), ([1, 83], None), ([3] * 300, None)):
    xh, nm, em = orc.random_inputs(n_list, 8, 3, pad)
    B = xh.shape[0]
    t = torch.full((B, 1), 0.4)
    with torch.no_grad():
        ref = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.")
        for prec in ("fp32", "bf16x6", "bf16x3"):
            dyn.precision = prec
            out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
            print(f"n={n_list[:4]}{'...' if len(n_list) > 4 else ''} N={xh.shape[1]} {prec}: rel {rel(out, ref):.1e} finite {bool(torch.isfinite(out).all())}")
# all-masked molecule inside a batch (node_mask row all False)
xh, nm, em = orc.random_inputs([4, 3], 8, 5, None)
nm2 = nm.clone(); nm2[1] = False; em2 = em.clone().view(2, 4, 4); em2[1] = False
xh2 = xh * nm2
with torch.no_grad():
    dyn.precision = "fp32"
    out = dyn._forward(torch.full((2, 1), 0.3).to(DEV), xh2.to(DEV), nm2.to(DEV), em2.view(em.shape).to(DEV), None, None).cpu()
print("batch with an empty molecule: finite", bool(torch.isfinite(out).all()), "empty rows zero", bool((out[1] == 0).all()))
# sampler: one molecule of one node, and a batch of single-node molecules
m = DiffusionQM9(default_config(hidden_nf=64, n_layers=2, timesteps=5))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}); m = m.to(DEV)
for n_list in ([1], [1, 1], [1, 7]):
    nm, em = orc.canonical_masks(n_list)
    x, h = m.sample_from_masks(nm.to(DEV), em.to(DEV), None)
    print("sample n =", n_list, "finite", bool(torch.isfinite(x).all() and torch.isfinite(h).all()), "single-node x == 0:", bool((x[0, 0].abs() < 1e-6).all()))
res = m.sample(3, DEV)
print("sample(3):", [tuple(r["x"].shape) for r in res])
res, names = m.sample_batches(2, 3, DEV)
print("sample_batches(2, 3):", len(res), names)
