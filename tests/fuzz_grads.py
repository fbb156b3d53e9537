"""Randomised gradient sweep of the training path: d(sum(out * w)) / d(every parameter, xh) of the HIP dynamics (hierdiff_amd.training) against
torch.autograd through the CPU oracle, on random model shapes / options / masks.  usage: fuzz_grads.py [cases] [seed] [big]
"big": widths 128 / 256 on batches of 20-36 molecules (where the whole-tile edge kernel runs and the forward keeps its
pre-activations), `training_precision` drawn from fp32 / fp16x3 and `keep_edge_activations` on / off (round 5)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from oracle import egnn_oracle as orc
from hierdiff_amd import EGNN_dynamics_QM9
from hierdiff_amd.weights import synthetic_state_dict
DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 3))
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
TOL = 1e-4
fails, worst_all = 0, 0.0
t0 = time.time()
for case in range(cases):
    H = int(rng.choice([32, 64, 128])); L = int(rng.integers(1, 3)); S = int(rng.integers(1, 4))
    att, tanh = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    C_ = int(rng.choice([0, 0, 1])); agg = str(rng.choice(["sum", "sum", "mean"]))
    nc = float(rng.choice([0.0, 1.0])); nf = float(rng.choice([1.0, 10.0]))
    B = int(rng.integers(1, 6)); nmax = int(rng.choice([3, 8, 17, 34]))
    n_list = [int(rng.integers(1, nmax + 1)) for _ in range(B)]
    tp, keep = "fp32", True
    if BIG:
        H = int(rng.choice([128, 256])); L = 1; S = int(rng.integers(1, 3))
        B = int(rng.integers(20, 37)); nmax = int(rng.choice([30, 34]))
        n_list = [int(rng.integers(nmax - 6, nmax + 1)) for _ in range(B)]
        tp = str(rng.choice(["fp32", "fp16x3", "fp16x3"])); keep = bool(rng.random() < 0.8)
    sd_np = synthetic_state_dict(9, C_, H, L, S, att, 5000 + case, 0.5)
    cfg = orc.DynCfg(in_node_nf=9, context_node_nf=C_, hidden_nf=H, n_layers=L, inv_sublayers=S, attention=att, tanh=tanh,
                     norm_constant=nc, normalization_factor=nf, aggregation_method=agg)
    xh, nm, em = orc.random_inputs(n_list, 8, 6000 + case, nmax + int(rng.integers(0, 2)))
    N = xh.shape[1]
    kind = int(rng.integers(0, 2))
    if kind == 1:
        emb = em.view(B, N, N).bool().clone()
        emb &= torch.from_numpy(rng.random((B, N, N)) > 0.25)
        em = emb.view(em.shape).to(em.dtype)
    ctx = torch.from_numpy(rng.standard_normal((B, N, C_)).astype(np.float32)) if C_ else None
    mol = None if rng.random() < 0.7 else int(rng.integers(1, N + 1))
    t = torch.from_numpy(rng.random((B, 1)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((B, N, 11)).astype(np.float32))
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in orc.as_torch_sd(sd_np).items()}
    xo = xh.clone().requires_grad_(True)
    ref = orc.dynamics_forward(sd, cfg, t, xo, nm, em, ctx, mol, prefix="dynamics.egnn.")
    (ref * w).sum().backward()
    dyn = EGNN_dynamics_QM9(9, C_, 3, hidden_nf=H, n_layers=L, attention=att, tanh=tanh, norm_constant=nc, inv_sublayers=S,
                            normalization_factor=nf, aggregation_method=agg)
    dyn.load_numpy_state_dict(sd_np, prefix="dynamics."); dyn = dyn.to(DEV); dyn.precision = "fp32"
    dyn.training_precision = tp; dyn.keep_edge_activations = keep
    xg = xh.to(DEV).requires_grad_(True)
    out = dyn._forward(t.to(DEV), xg, nm.to(DEV), em.to(DEV), None if ctx is None else ctx.to(DEV), mol)
    (out * w.to(DEV)).sum().backward()
    scale = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    worst, bad = 0.0, []
    for name, p in dyn.egnn.named_parameters():
        r = sd["dynamics.egnn." + name].grad
        if r is None:           # e.g. att_mlp of a model without attention is not in the oracle's graph
            continue
        r = r.double().numpy(); g = p.grad.detach().cpu().double().numpy()
        err = np.linalg.norm(g - r); bound = TOL * np.linalg.norm(r) + 1e-7 * scale * np.sqrt(r.size)
        worst = max(worst, err / max(np.linalg.norm(r), 1e-30))
        if not err <= bound: bad.append(name)
    valid = nm.numpy()[..., 0]
    gx, rx = xg.grad.cpu().double().numpy(), xo.grad.double().numpy()
    ex = np.linalg.norm(gx[valid] - rx[valid]) / max(np.linalg.norm(rx[valid]), 1e-30)
    if ex > TOL or np.any(gx[~valid] != 0.0): bad.append("xh")
    vr = float((out.detach().cpu().double() - ref.detach().double()).norm() / max(float(ref.detach().double().norm()), 1e-30))
    if vr > 1e-5: bad.append("value")
    fails += int(bool(bad)); worst_all = max(worst_all, worst)
    print(f"case {case:3d} H={H:3d} L={L} S={S} att={int(att)} tanh={int(tanh)} C={C_} agg={agg:4s} nc={nc} nf={nf:4.1f} "
          f"n={n_list if not BIG else str(B) + ' x ' + str(min(n_list)) + '..' + str(max(n_list))} {tp if BIG else ''}{'' if keep else ' recompute'} mask={kind} mol={mol}"
          f"  value {vr:.1e} worst grad {worst:.1e} d/dxh {ex:.1e}{' FAIL ' + ','.join(bad) if bad else ''}", flush=True)
print(f"{cases} cases in {time.time() - t0:.0f} s, failures {fails}, worst parameter-gradient rel-L2 {worst_all:.2e}")
sys.exit(1 if fails else 0)
