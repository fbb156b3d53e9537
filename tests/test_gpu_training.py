"""GPU tier (-m gpu): training forward / backward of the EGNN dynamics.

Gradient parity: every parameter gradient (and the input gradient) of the HIP path - `hd_edge_layer_forward` /
`hd_edge_layer_backward` under hierdiff_amd.training's autograd Function, node-level GEMMs on the library's own
exact-fp32 GEMM `hd_gemm_f32` (forward, dX, split-K dW with the bias gradient) - against torch.autograd of the CPU oracle on the same inputs.  Bar: rel-L2 < 1e-4 per tensor (exact-fp32 kernels).
"""
import numpy as np
import pytest
import torch

from oracle import egnn_oracle as orc
from tests.helpers import fixture_model, load, rel_l2
from tests.test_gpu_parity import DEV, build_diffusion, build_dynamics

pytestmark = [pytest.mark.gpu, pytest.mark.autograd]

GRAD_TOL = 1e-4


def _oracle_sd(sd_np):
    return {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in orc.as_torch_sd(sd_np).items()}


def _compare_grads(named_params, sd, prefix, what, tol=GRAD_TOL, skip=()):
    worst, n = 0.0, 0
    scale = max(float(v.grad.abs().max()) for k, v in sd.items() if v.grad is not None)
    for name, p in named_params:
        key = prefix + name
        if any(s in key for s in skip):
            continue
        ref = sd[key].grad
        assert ref is not None, key
        assert p.grad is not None, f"{what}: no gradient for {key}"
        got = p.grad.detach().cpu().double().numpy()
        r = ref.double().numpy()
        err = np.linalg.norm(got - r)
        bound = tol * np.linalg.norm(r) + 1e-7 * scale * np.sqrt(r.size)     # tensors whose gradient is ~0 compare absolutely
        assert err <= bound, f"{what}: grad of {key}: |diff| {err:.3e} > {bound:.3e} (|ref| {np.linalg.norm(r):.3e})"
        worst = max(worst, err / max(np.linalg.norm(r), 1e-30))
        n += 1
    return worst, n


CASES = [
    # n_list, H, L, C, n_max, mol_shape, general_mask
    ([5, 3, 4], 32, 2, 0, None, None, False),
    ([1, 2, 2, 7, 33], 32, 1, 0, None, None, False),            # single node, one-edge segments, 2-tile segments
    ([9, 6, 12], 64, 2, 1, 12, 9, True),                        # context, fixed trailing nodes, holes / self edge in the mask
    ([30, 30, 17, 30], 256, 2, 0, None, None, False),           # production width
]


@pytest.mark.parametrize("n_list,H,L,C_,n_max,mol,general", CASES)
def test_dynamics_value_and_gradients_vs_oracle_autograd(n_list, H, L, C_, n_max, mol, general):
    from hierdiff_amd.weights import synthetic_state_dict
    sd_np = synthetic_state_dict(9, C_, H, L, 2, True, 60 + H, 0.5)
    cfg = orc.DynCfg(in_node_nf=9, context_node_nf=C_, hidden_nf=H, n_layers=L, normalization_factor=10.0)
    xh, nm, em = orc.random_inputs(n_list, 8, 71, n_max)
    B, N = xh.shape[:2]
    if general:
        em = em.clone()
        em[0, :4, 4:9] = False; em[0, 4:9, :4] = False
        em[1, 2, 2] = True
        em[2, 0, 1] = False
    t = torch.linspace(0.1, 0.9, B).view(B, 1)
    ctx = (torch.zeros(B, N, C_) + torch.linspace(0.5, 2.0, B).view(B, 1, 1)) if C_ else None
    g = torch.Generator().manual_seed(5)
    w = torch.randn(B, N, 11, generator=g)
    # oracle + autograd
    sd = _oracle_sd(sd_np)
    xo = xh.clone().requires_grad_(True)
    ref = orc.dynamics_forward(sd, cfg, t, xo, nm, em, ctx, mol, prefix="dynamics.egnn.")
    (ref * w).sum().backward()
    # HIP path
    dyn = build_dynamics(sd_np, H, L, C_=C_)
    dyn.precision = "fp32"
    xg = xh.to(DEV).requires_grad_(True)
    out = dyn._forward(t.to(DEV), xg, nm.to(DEV), em.to(DEV), None if ctx is None else ctx.to(DEV), mol)     # autograd recording: differentiable path
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    with torch.no_grad():       # the differentiable forward equals the sampler's forward
        inf = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None if ctx is None else ctx.to(DEV), mol)
    assert rel_l2(out.detach().cpu().numpy(), inf.cpu().numpy()) < 2e-6
    (out * w.to(DEV)).sum().backward()
    worst, n = _compare_grads(dyn.egnn.named_parameters(), sd, "dynamics.egnn.", f"H={H} L={L}")
    gx = xg.grad.cpu().double().numpy()
    rx = xo.grad.double().numpy()
    valid = nm.numpy()[..., 0]
    assert rel_l2(gx[valid], rx[valid]) < GRAD_TOL, "input gradient"
    assert np.all(gx[~valid] == 0.0)
    print(f"n={n_list} H={H} L={L}: {n} parameter tensors, worst grad rel-L2 {worst:.2e}, d/dxh {rel_l2(gx[valid], rx[valid]):.2e}")


@pytest.mark.parametrize("name", ["f9_nll_train_h64_l2", "f9_nll_eval_h64_l2"])
def test_training_loss_gradients_on_reference_fixtures(name):
    """F9 (the reference's own compute_loss draws): d(mean loss)/d(every dynamics parameter) of DiffusionQM9.compute_loss
    on the HIP path vs autograd through the oracle's nll_forward, schedule values replayed on both sides; the loss value
    itself is pinned to the reference by the fixture."""
    fx = load(name)
    sd_np, _, cfg = fixture_model(fx)
    training = bool(int(fx["training"]))
    T = int(fx["T"])
    nm, em = orc.canonical_masks([int(v) for v in fx["n_list"]])
    gam = {k: torch.from_numpy(fx[k]) for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
    sd = _oracle_sd(sd_np)
    ref, _ = orc.nll_forward(sd, cfg, T, fx["x"], fx["h"], nm, em, None, fx["t_int"], fx["eps"],
                             None if training else fx["eps0"], training=training, gammas=gam)
    ref.mean().backward()
    model = build_diffusion(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), T=T, precision="fp32")
    model.train(training)
    model.dynamics.differentiable = True          # gradients of the EVALUATION-mode loss too (by rule only training mode records)
    replay = dict(t_int=fx["t_int"], eps=fx["eps"], gammas={k: fx[k] for k in gam})
    if not training:
        replay["eps0"] = fx["eps0"]
    loss, _ = model.compute_loss(torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["h"]).to(DEV), nm.to(DEV), em.to(DEV),
                                 None, t0_always=not training, **replay)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), fx["loss"], rtol=1e-4, atol=1e-3)
    loss.mean().backward()
    worst, n = _compare_grads(model.dynamics.egnn.named_parameters(), sd, "dynamics.egnn.", name)
    print(f"{name}: {n} parameter tensors, worst grad rel-L2 {worst:.2e}")


def _fused_calls(monkeypatch):
    """Counts the launches of the fused loss (hierdiff_amd.training.vlb_loss) so a test can assert WHICH path produced its numbers."""
    import hierdiff_amd.training as tr
    calls = []
    real = tr.vlb_loss

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)
    monkeypatch.setattr(tr, "vlb_loss", spy)
    return calls


@pytest.mark.parametrize("case", ["f12_l2_poly2", "f20_norm_values", "f13_elem", "f9_vlb_batch_entry"])
def test_fused_training_loss_meets_reference_fixtures(case, monkeypatch):
    """Round 6: the fused training loss (csrc/k_loss.hpp, the default training path on the GPU) against
    numbers the REFERENCE produced in training mode, under autograd, branch by branch: `l2` + predefined schedule with a t = 0 row
    (F12), non-unit norm_values / norm_biases through `nll` (F20), 3 + 0 `elem` feature columns (F13, training-mode loss added to
    the generator this round), and the vlb / learned-schedule case through the batch-level entry `forward(batch)` (F9).  Each case
    asserts that the fused kernel - not the torch-op path - produced the value, and that the value is differentiable."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    calls = _fused_calls(monkeypatch)
    if case == "f12_l2_poly2":
        fx = load("f12_poly2_l2_h32_l2")
        H, L, T = int(fx["hidden_nf"]), int(fx["n_layers"]), int(fx["T"])
        cfg = default_config(hidden_nf=H, n_layers=L, timesteps=T)
        cfg["noise_schedule"], cfg["loss_type"] = "polynomial_2", "l2"
        cfg["pre_noise"] = dict(noise_schedule="polynomial_2", timesteps=T, precision=1e-4)
        model = DiffusionQM9(cfg)
        sd_np = {k: v for k, v in synthetic_state_dict(9, 0, H, L, 2, True, int(fx["weight_seed"]), 1.0).items() if not k.startswith("gamma.")}
        sd_np["gamma.gamma"] = fx["gamma_table"]
        model.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()})
        model = model.to(DEV).train()
        nm, em = orc.canonical_masks([int(v) for v in fx["n_list"]])
        assert float(fx["t_int"][0, 0]) == 0.0
        loss, info = model.compute_loss(torch.from_numpy(fx["loss_x"]).to(DEV), torch.from_numpy(fx["loss_h"]).to(DEV), nm.to(DEV),
                                        em.to(DEV), None, t0_always=False, t_int=fx["t_int"], eps=fx["eps"])
        np.testing.assert_allclose(loss.detach().cpu().numpy(), fx["loss"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(info["error"].detach().cpu().numpy(), fx["error"], rtol=1e-4, atol=1e-5)
    elif case == "f20_norm_values":
        fx = load("f20_norm_h64_l2")
        sd_np, _, _ = fixture_model(fx)
        cfg = default_config(hidden_nf=int(fx["hidden_nf"]), n_layers=int(fx["n_layers"]), timesteps=int(fx["T"]))
        cfg.norm_values = [float(v) for v in fx["norm_values"]]
        cfg.norm_biases = [None] + [float(v) for v in fx["norm_biases"][1:]]
        model = DiffusionQM9(cfg)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()})
        model = model.to(DEV).train()
        nm, em = orc.canonical_masks([int(v) for v in fx["n_list"]])
        gam = {k: fx[f"train_{k}"] for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
        loss = model.nll(torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["h"]).to(DEV), nm.to(DEV), em.to(DEV), None,
                         t_int=fx["train_t_int"], eps=fx["train_eps"], gammas=gam)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), fx["train_nll"], rtol=1e-4, atol=1e-3)
    elif case == "f13_elem":
        fx = load("f13_elem_h64_l2")
        H, L = int(fx["hidden_nf"]), int(fx["n_layers"])
        cfg = default_config(hidden_nf=H, n_layers=L, timesteps=1000)
        cfg["node_coarse_type"] = "elem"
        model = DiffusionQM9(cfg)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(4, 0, H, L, 2, True, int(fx["weight_seed"]), 1.0).items()})
        model = model.to(DEV).train()
        nm, em = torch.from_numpy(fx["node_mask"]), torch.from_numpy(fx["edge_mask"])
        gam = {k: fx[f"train_{k}"] for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
        assert float(fx["train_t_int"][0, 0]) == 0.0
        loss, info = model.compute_loss(torch.from_numpy(fx["loss_x"]).to(DEV), torch.from_numpy(fx["loss_h"]).to(DEV), nm.to(DEV),
                                        em.to(DEV), None, t0_always=False, t_int=fx["train_t_int"], eps=fx["train_eps"], gammas=gam)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), fx["train_loss"], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(info["error"].detach().cpu().numpy(), fx["train_error"], rtol=1e-4, atol=1e-4)
    else:
        fx = load("f9_nll_train_h64_l2")
        sd_np, _, _ = fixture_model(fx)
        model = build_diffusion(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), T=int(fx["T"]), precision="fp32").train()
        nm, em = orc.canonical_masks([int(v) for v in fx["n_list"]])
        x, h = torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["h"]).to(DEV)
        B, N = x.shape[:2]
        batch = {"positions": x, "atom_mask": nm.to(DEV), "edge_mask": em.to(DEV).view(B, N, N), "node_feature": h}
        gam = {k: fx[k] for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
        loss = model(batch, t_int=fx["t_int"], eps=fx["eps"], gammas=gam)["loss"]
        assert abs(loss.item() - float(np.mean(fx["loss"]))) <= 1e-4 * abs(float(np.mean(fx["loss"]))) + 1e-3
    assert len(calls) == 1, "the value above must come from the fused loss kernel"
    assert loss.requires_grad
    loss.mean().backward()
    grads = [p.grad for p in model.dynamics.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(g).all() for g in grads) and any(float(g.abs().max()) > 0 for g in grads)


@pytest.mark.parametrize("loss_type,schedule,coarse,nv", [("vlb", "learned", "prop", (1.0, 1.0, 1.0)), ("l2", "polynomial_2", "prop", (1.0, 1.0, 1.0)),
                                                          ("vlb", "polynomial_2", "prop", (1.0, 1.0, 1.0)), ("vlb", "learned", "elem", (1.0, 1.0, 1.0)),
                                                          ("vlb", "learned", "prop", (2.0, 4.0, 1.0))])
def test_fused_training_loss_equals_the_torch_op_path(loss_type, schedule, coarse, nv):
    """Round 5: compute_loss in training mode as two fused launches per direction (csrc/k_loss.hpp: z_t, then everything behind the
    network call) against the torch-op path it replaces (`model.fused_loss = False`, itself pinned to the reference by fixtures F9)
    on the same draws: per-molecule loss values, the `error` entry, and the gradient of the mean loss with respect to EVERY
    parameter - dynamics and schedule network - over the config branches the kernel carries (vlb / l2, learned / fixed schedule,
    5 + 3 / 3 + 0 feature columns, non-unit norm_values), ragged molecules, rows with t = 0 (the integer likelihood) among the others."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, B, N = 64, 2, 7, 11
    cfg = default_config(hidden_nf=H, n_layers=L, timesteps=50)
    cfg.loss_type, cfg.noise_schedule, cfg.node_coarse_type = loss_type, schedule, coarse
    cfg.pre_noise.noise_schedule = schedule
    cfg.norm_values = list(nv)
    g = torch.Generator().manual_seed(3)
    sizes = [11, 7, 1, 9, 4, 11, 2]
    nm = torch.zeros(B, N, 1, dtype=torch.bool)
    for b, n in enumerate(sizes):
        nm[b, :n] = True
    em = (nm.float() @ nm.float().transpose(1, 2)).bool() & ~torch.eye(N, dtype=torch.bool)[None]
    x = torch.randn(B, N, 3, generator=g) * nm
    x = x - (x.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
    F_ = 8 if coarse == "prop" else 3
    hi = torch.randint(0, 5, (B, N, 5 if coarse == "prop" else 3), generator=g).float()
    h = (torch.cat([hi, torch.randn(B, N, 3, generator=g)], 2) if coarse == "prop" else hi) * nm
    batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.to(DEV), "node_feature": h.to(DEV)}
    t_int = torch.tensor([[0.], [17.], [3.], [0.], [50.], [1.], [29.]])
    res = {}
    for fused in (True, False):
        torch.manual_seed(11)
        m = DiffusionQM9(cfg)
        sd = synthetic_state_dict(3 + F_ + 1 - 3, 0, H, L, 2, True, 41, 0.5)
        own = m.state_dict()
        m.load_state_dict({k: (torch.from_numpy(sd[k].copy()) if k in sd and tuple(sd[k].shape) == tuple(v.shape) else v) for k, v in own.items()})
        m = m.to(DEV).train()
        m.fused_loss = fused
        eps = m.sample_combined_position_feature_noise(B, N, nm.to(DEV)) if "eps" not in res else res["eps"]
        res["eps"] = eps
        xs, hs, dl = m.normalize(batch["positions"], batch["node_feature"], nm.to(DEV).float())
        per, info = m.compute_loss(xs, hs, nm.to(DEV), em.reshape(B, N * N).to(DEV), None, t0_always=False, t_int=t_int, eps=eps)
        out = m.forward(batch, t_int=t_int, eps=eps)["loss"]
        out.backward()
        res[fused] = (per.detach().clone(), info["error"].detach().clone(), out.detach().clone(),
                      {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        # the four schedule values as leaves: their gradients element by element (the schedule network's parameter gradients are
        # sums of these 1e4-1e5-sized numbers with opposite signs - one rounding of a term is visible in such a sum)
        with torch.no_grad():
            tt = t_int.to(DEV)
            gv = m.gamma(torch.cat([(tt - 1) / m.T, tt / m.T, torch.zeros_like(tt), torch.ones_like(tt)], 0)).view(4, B, 1)
        leaves = {k: gv[i].clone().requires_grad_(True) for i, k in enumerate(("gamma_s", "gamma_t", "gamma_0", "gamma_T"))}
        m.zero_grad(set_to_none=True)
        per2, _ = m.compute_loss(xs, hs, nm.to(DEV), em.reshape(B, N * N).to(DEV), None, t0_always=False, t_int=t_int, eps=eps, gammas=leaves)
        per2.mean().backward()
        res[("dgam", fused)] = torch.stack([(leaves[k].grad if leaves[k].grad is not None else torch.zeros(B, 1, device=DEV)).view(B)
                                            for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")])      # (l2: g_s, g_0 do not enter)
    dga, dgb = res[("dgam", True)], res[("dgam", False)]
    np.testing.assert_allclose(dga.cpu().numpy(), dgb.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(dgb.abs().max()))
    scale_g = float(dgb.abs().max())
    a, b = res[True], res[False]
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[2])
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(float(a[2]), float(b[2]), rtol=2e-5)
    assert set(a[3]) == set(b[3]) and len(a[3]) > 40
    scale = max(float(v.abs().max()) for v in b[3].values())
    worst = 0.0
    for k, gb in b[3].items():
        err = float((a[3][k] - gb).norm())
        # (the small schedule-network gradients - l3.bias is the plain sum of all 4 B schedule-value gradients, ~0 against terms of
        # ~1e3 - are rounding noise of those terms in BOTH paths: an absolute bar for them; the tight check is `dgam` above)
        slack = 2e-5 * scale_g * gb.numel() ** 0.5 if k.startswith("gamma.") else 1e-7 * scale * gb.numel() ** 0.5
        assert err <= 1e-4 * float(gb.norm()) + slack, (k, err, float(gb.norm()))
        if not k.startswith("gamma."):
            worst = max(worst, err / max(float(gb.norm()), 1e-30))
    sched = [k for k in a[3] if k.startswith("gamma.")]
    assert (schedule == "learned") == bool(sched)
    print(f"{loss_type} {schedule} {coarse} nv={nv}: loss {float(a[2]):.4f} / {float(b[2]):.4f}, {len(a[3])} gradients ({len(sched)} of the schedule network), "
          f"worst dynamics-gradient rel-L2 fused vs torch ops {worst:.2e}; d loss / d gamma values up to {scale_g:.1e}, fused vs torch ops within 1e-4")


def test_training_step_with_learned_schedule_and_optimizer():
    """training_step (diffusion_qm9.py:774-777) end to end: learned schedule in the graph, every parameter gets a finite
    gradient; the loss code's gradient with respect to the schedule values equals the oracle's; a few Adam steps reduce
    the loss on a fixed batch."""
    from hierdiff_amd.weights import synthetic_state_dict
    H, L = 32, 2
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 33, 0.5)
    n_list = [6, 4, 7, 5]
    nm, em = orc.canonical_masks(n_list)
    B, N = nm.shape[:2]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, N, 3, generator=g) * nm
    x = x - (x.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
    h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], dim=2) * nm
    t_int = torch.tensor([[0.], [17.], [500.], [1000.]])
    eps = orc.combined_noise(torch.randn(B, N, 3, generator=g), torch.randn(B, N, 8, generator=g), nm.float())
    model = build_diffusion(sd_np, H, L, precision="fp32")
    model.train(True)
    batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.to(DEV), "node_feature": h.to(DEV)}
    loss = model.forward(batch, t_int=t_int, eps=eps)["loss"]
    loss.backward()
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
    sd = _oracle_sd(sd_np)
    cfg = orc.DynCfg(hidden_nf=H, n_layers=L)
    ref, _ = orc.nll_forward(sd, cfg, 1000, x, h, nm, em, None, t_int, eps, None, training=True)
    ref.mean().backward()
    # learned schedule evaluated in fp32 on both sides: gamma differs by ~1e-4 between hosts and the SNR weight
    # exp(gamma_t - gamma_s) - 1 by ~1 %, so this end-to-end comparison is loose by construction
    assert abs(loss.item() - ref.mean().item()) <= 3e-2 * abs(ref.mean().item())
    # (the gradients inherit that spread - rows are weighted by the SNR factor - so they are compared below, where both
    # sides are fed the same schedule values)
    # the sharp check of the schedule path: d(loss)/d(gamma_s, gamma_t, gamma_0, gamma_T) with the SAME gamma values as
    # leaves on both sides (what the gamma network's own backward - plain torch autograd - is fed with)
    with torch.no_grad():
        tt = t_int / 1000
        vals = {"gamma_s": model.gamma(((t_int - 1) / 1000).to(DEV)).cpu(), "gamma_t": model.gamma(tt.to(DEV)).cpu(),
                "gamma_0": model.gamma(torch.zeros(B, 1, device=DEV)).cpu(), "gamma_T": model.gamma(torch.ones(B, 1, device=DEV)).cpu()}
    leaf_g = {k: v.clone().to(DEV).requires_grad_(True) for k, v in vals.items()}
    leaf_c = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
    model.zero_grad()
    lg, _ = model.compute_loss(x.to(DEV), h.to(DEV), nm.to(DEV), em.to(DEV), None, t0_always=False, t_int=t_int, eps=eps, gammas=leaf_g)
    lg.mean().backward()
    sd2 = _oracle_sd(sd_np)
    lc, _ = orc.nll_forward(sd2, cfg, 1000, x, h, nm, em, None, t_int, eps, None, training=True, gammas=leaf_c)
    lc.mean().backward()
    np.testing.assert_allclose(lg.detach().cpu().numpy(), lc.detach().numpy(), rtol=1e-4, atol=1e-3)
    for k in vals:
        assert rel_l2(leaf_g[k].grad.cpu().numpy(), leaf_c[k].grad.numpy()) < GRAD_TOL, k
    _compare_grads(model.dynamics.egnn.named_parameters(), sd2, "dynamics.egnn.", "schedule leaves: dynamics")
    # a few optimiser steps on the fixed batch
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    first = last = None
    for it in range(8):
        opt.zero_grad()
        loss = model.forward(batch, t_int=t_int, eps=eps)["loss"]
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
        last = loss.item()
    assert np.isfinite(last) and last < first, (first, last)
    loss = model.training_step(batch, 0)            # the un-replayed entry point (own draws)
    assert torch.isfinite(loss) and loss.requires_grad


@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_eval_mode_nll_does_not_depend_on_grad_mode(precision):
    """an evaluation call made WITHOUT torch.no_grad() (this test runs with autograd recording) takes the
    inference kernels of the configured precision - it neither raises in the fp16x3 mode nor switches the schedule from the
    float64 table to the on-device fp32 network - and returns the bits of the same call under no_grad."""
    from hierdiff_amd.weights import synthetic_state_dict
    H, L = 64, 2
    model = build_diffusion(synthetic_state_dict(9, 0, H, L, 2, True, 34, 0.5), H, L, T=1000, precision=precision).eval()
    assert torch.is_grad_enabled() and any(p.requires_grad for p in model.dynamics.parameters())
    nm, em = orc.canonical_masks([6, 4, 7, 5])
    B, N = nm.shape[:2]
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, N, 3, generator=g) * nm
    x = x - (x.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
    h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], dim=2) * nm
    replay = dict(t_int=torch.tensor([[3.], [170.], [500.], [1000.]]), eps=torch.randn(B, N, 11, generator=g) * nm,
                  eps0=torch.randn(B, N, 11, generator=g) * nm)
    args = (x.to(DEV), h.to(DEV), nm.to(DEV), em.to(DEV))
    with_grad = model.nll(*args, **replay)
    assert not with_grad.requires_grad
    with torch.no_grad():
        without = model.nll(*args, **replay)
    assert torch.isfinite(with_grad).all() and torch.equal(with_grad, without)


# ----------------------------------------------------------------------------- hd_gemm_f32 (csrc/k_tgemm.hpp)

GEMM_SHAPES = [(7680, 256, 512), (70, 9, 10), (300, 130, 77), (1, 256, 256), (129, 512, 256), (64, 128, 16), (33, 10, 256),
               (7680, 512, 256), (4101, 256, 256)]        # (round 6: + the row-resident kernel's other column width and a ragged row count)


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_f32_three_layouts_vs_float64(M, N, K):
    """The training GEMM in its three operand layouts - Y = X W^T + b, dX = dY W, dW = dY^T X with the bias gradient as
    column sums (split-K) - against a float64 product of the same operands: rel-L2 < 2e-6 (exact fp32 products, fp32
    accumulation over K), on aligned production shapes and on odd / unaligned ones (scalar-load path, partial tiles, partial
    K chunks, K smaller than a chunk), with non-unit row strides (views into wider matrices)."""
    from hierdiff_amd import training as tr
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    X = torch.randn(M, K + 3, generator=g).to(DEV)[:, :K]              # row stride K + 3: unaligned rows unless (K + 3) % 4 == 0
    W = torch.randn(N, K + 2, generator=g).to(DEV)[:, 1:K + 1]          # offset start: unaligned base
    b = torch.randn(N, generator=g).to(DEV)
    Xc, Wc = X.contiguous(), W.contiguous()
    ref = Xc.double() @ Wc.double().t() + b.double()
    for xx, ww in ((X, W), (Xc, Wc)):
        y = tr._linear_fwd(xx, ww, b)
        assert rel_l2(y.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    pre, act = tr._linear_fwd(Xc, Wc, b, tr._EPI_BIAS_SILU2)
    assert rel_l2(pre.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert rel_l2(act.cpu().numpy(), torch.nn.functional.silu(ref).cpu().numpy()) < 2e-6
    gy = torch.randn(M, N, generator=g).to(DEV)
    dx = tr._linear_dx(gy, Wc)
    assert rel_l2(dx.cpu().numpy(), (gy.double() @ Wc.double()).cpu().numpy()) < 2e-6
    dx2 = tr._linear_dx(gy, W)                                          # strided weight rows
    assert rel_l2(dx2.cpu().numpy(), (gy.double() @ Wc.double()).cpu().numpy()) < 2e-6
    pre_k = torch.randn(M, K, generator=g).to(DEV)
    dpre = tr._linear_dx(gy, Wc, tr._EPI_MUL_DSILU, aux=pre_k)
    sg = torch.sigmoid(pre_k.double())
    ref_d = (gy.double() @ Wc.double()) * (sg * (1 + pre_k.double() * (1 - sg)))
    assert rel_l2(dpre.cpu().numpy(), ref_d.cpu().numpy()) < 2e-6
    dW, db = tr._linear_dw(gy, Xc, True)
    assert rel_l2(dW.cpu().numpy(), (gy.double().t() @ Xc.double()).cpu().numpy()) < 2e-6
    assert rel_l2(db.cpu().numpy(), gy.double().sum(0).cpu().numpy()) < 2e-6
    dW2, none = tr._linear_dw(gy, X, False, rows=max(1, M - 5))          # a row prefix, strided x, no bias gradient
    assert none is None
    r = max(1, M - 5)
    assert rel_l2(dW2.cpu().numpy(), (gy[:r].double().t() @ Xc[:r].double()).cpu().numpy()) < 2e-6
    # residual + row-mask epilogue of the node MLP
    res = torch.randn(M, N, generator=g).to(DEV)
    mask = (torch.rand(M, generator=g) > 0.3).float().to(DEV)
    out = torch.empty(M, N, device=DEV)
    tr._gemm(M, N, K, Xc, Xc.stride(0), 1, Wc, 1, Wc.stride(0), out, bias=b, epi=tr._EPI_RESID_MASK, aux=res, rmask=mask)
    assert rel_l2(out.cpu().numpy(), ((res.double() + ref) * mask.double()[:, None]).cpu().numpy()) < 2e-6
    # deterministic: the split-K sum runs in slab order
    dW_b, db_b = tr._linear_dw(gy, Xc, True)
    assert torch.equal(dW, dW_b) and torch.equal(db, db_b)


def test_gemm_f32_rejects_bad_arguments():
    from hierdiff_amd import _lib
    lib = _lib.load()
    a = torch.zeros(8, 8, device=DEV)
    s = 0
    # no unit stride on A
    assert lib.hd_gemm_f32(0, 8, 8, 8, a.data_ptr(), 8, 2, a.data_ptr(), 1, 8, a.data_ptr(), 8, None, 0, None, None, None, 1, None, None, s) < 0
    # split-K without a workspace
    assert lib.hd_gemm_f32(0, 8, 8, 8, a.data_ptr(), 8, 1, a.data_ptr(), 1, 8, a.data_ptr(), 8, None, 0, None, None, None, 4, None, None, s) < 0
    # SiLU epilogue without its second output
    assert lib.hd_gemm_f32(0, 8, 8, 8, a.data_ptr(), 8, 1, a.data_ptr(), 1, 8, a.data_ptr(), 8, None, 1, None, None, None, 1, None, None, s) < 0
    assert b"hd_gemm_f32" in lib.hd_last_error()


def test_training_step_launches_no_blas_library_kernel():
    """A training step (loss forward + backward) runs no GEMM of the BLAS library: torch's profiler sees none of its
    kernel families (Cijk_* = hipBLASLt / Tensile, rocblas_*) among the device kernels."""
    from torch.profiler import ProfilerActivity, profile
    from hierdiff_amd.weights import synthetic_state_dict
    H, L = 64, 2
    m = build_diffusion(synthetic_state_dict(9, 0, H, L, 2, True, 5, 0.5), H, L, T=50).train()
    torch.manual_seed(0)
    B, N = 6, 12
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, N, 3, generator=g)
    x = x - x.mean(1, keepdim=True)
    h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
    batch = {"positions": x.to(DEV), "atom_mask": torch.ones(B, N, 1, dtype=torch.bool, device=DEV),
             "edge_mask": (~torch.eye(N, dtype=torch.bool))[None].expand(B, N, N).contiguous().to(DEV), "node_feature": h.to(DEV)}

    def step():
        for p in m.parameters():
            p.grad = None
        m.training_step(batch, 0).backward()

    step()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step()
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages() if getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)]
    assert any("k_tgemm" in n for n in names), names[:40]
    bad = [n for n in names if n.startswith("Cijk_") or "rocblas" in n.lower() or "hipblas" in n.lower()]
    assert not bad, bad


def test_colsum_f32_matches_a_float64_column_sum_and_is_deterministic():
    """hd_colsum_f32 (the per-tile partial sums of hd_edge_layer_backward -> db2, d(w_r)/d(w_d), d(wa), d(ba)): up to four arrays of
    different widths in one call, against float64 column sums; same bits on every call; ragged row counts."""
    import ctypes as C
    from hierdiff_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    for rows in (1, 31, 6976):
        arrs = [torch.randn(rows, w, generator=g).to(DEV) for w in (256, 512, 256, 1)]
        for n in (4, 3, 1):
            outs = []
            for rep in range(2):
                dst = [torch.full((a.shape[1],), float("nan"), device=DEV) for a in arrs[:n]]
                ws = torch.empty(32 * sum(a.shape[1] for a in arrs[:n]), device=DEV)
                _lib.check(lib.hd_colsum_f32(0, rows, n, (C.c_void_p * n)(*[a.data_ptr() for a in arrs[:n]]),
                                             (C.c_int * n)(*[a.shape[1] for a in arrs[:n]]),
                                             (C.c_void_p * n)(*[d.data_ptr() for d in dst]), ws.data_ptr(), 0), "hd_colsum_f32")
                torch.cuda.synchronize()
                outs.append(dst)
            for a, d0, d1 in zip(arrs[:n], outs[0], outs[1]):
                ref = a.double().sum(0)
                assert float((d0.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())) * max(1, rows) ** 0.5
                assert torch.equal(d0, d1)


def test_randomised_gradient_sweep():
    """The wide net next to the fixed cases: tests/fuzz_grads.py on random shapes / options / masks (a 120-case run of the same
    script: profiles/r03_fuzz_parity.log)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_grads.py"), "16", "9"], cwd=root,
                          capture_output=True, text=True, timeout=600)
    tail = "\n".join(proc.stdout.splitlines()[-4:])
    assert proc.returncode == 0, tail + proc.stderr[-2000:]
    assert "failures 0" in tail, tail
    # round 5: widths 128 / 256 on 20-36 molecules - kept pre-activations, training_precision fp32 / fp16x3 drawn per case
    # (a 24-case run: profiles/r05_fuzz_grads_big.log)
    proc = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_grads.py"), "8", "13", "big"], cwd=root,
                          capture_output=True, text=True, timeout=600)
    tail = "\n".join(proc.stdout.splitlines()[-4:])
    assert proc.returncode == 0, tail + proc.stderr[-2000:]
    assert "failures 0" in tail, tail


def test_trainer_ddp_step_equals_the_manual_sequence():
    """hierdiff_amd.trainer.ddp_step on the real model (world size 1: no process group, the all-reduce is skipped): forward + backward
    on the HIP path, clip_grad_norm_(2), AdamW with the reference's values - bit-equal to the same sequence written out."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.trainer import configure_optimizers, ddp_step
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, B, N = 64, 2, 6, 9
    sd = synthetic_state_dict(9, 0, H, L, 2, True, 31, 0.5)
    g = torch.Generator().manual_seed(2)
    nm = torch.ones(B, N, 1, dtype=torch.bool); nm[2, 5:] = False; nm[4, 3:] = False
    em = (nm.float() @ nm.float().transpose(1, 2)).bool() & ~torch.eye(N, dtype=torch.bool)[None]
    x = torch.randn(B, N, 3, generator=g) * nm
    x = x - (x.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
    h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2) * nm
    batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.to(DEV), "node_feature": h.to(DEV)}

    def fresh():
        m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        return m.to(DEV).train()
    a, b = fresh(), fresh()
    opt_a, _ = configure_optimizers(a)
    opt_b, _ = configure_optimizers(b)
    for step in range(2):
        torch.manual_seed(100 + step)                       # the loss draws t and eps from torch's generator
        out = ddp_step(a, batch, opt_a, clip_val=2.0)
        torch.manual_seed(100 + step)
        opt_b.zero_grad(set_to_none=True)
        loss = b.training_step(batch, 0)
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_([p for p in b.parameters() if p.grad is not None], 2.0)
        opt_b.step()
        assert out["loss"] == float(loss.detach()) and out["grad_norm"] == float(norm) and np.isfinite(out["loss"])
    for (ka, pa), (kb, pb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(pa, pb), ka
    moved = sum(float((pa.cpu() - torch.from_numpy(sd[k])).abs().max()) > 0 for k, pa in a.state_dict().items() if k in sd and k != "buffer")
    assert moved > 10, "the optimiser must have moved the weights"


def test_inference_path_sees_the_weights_a_fused_optimizer_wrote():
    """torch's fused optimizers (the AdamW `trainer.configure_optimizers` picks on the GPU) change the parameters without bumping their
    version counters, which the packed weight images of the inference handle and the schedule table are cached against: the cache
    keys also carry a count of optimizer steps (`_lib.optimizer_generation`).  After a fused step the no-grad forward (packed images,
    HIP sampler kernels) must equal the differentiable forward (the parameters themselves), and the tabulated schedule the network."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.trainer import configure_optimizers
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, B, N = 64, 2, 5, 9
    m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L, timesteps=50))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 33, 0.5).items()})
    m = m.to(DEV).train()
    batch = {k: v.to(DEV) for k, v in _host_batch(5, B, N).items()}
    opt, _ = configure_optimizers(m, lr=3e-2)
    assert opt.defaults.get("fused")
    xh = torch.randn(B, N, 11, device=DEV) * batch["atom_mask"]
    t = torch.full((B, 1), 0.5, device=DEV)
    em = batch["edge_mask"].reshape(B, N * N)

    def both():
        with torch.no_grad():
            inf = m.dynamics._forward(t, xh, batch["atom_mask"], em, None, None)
            tab = m._gamma_rows(t, "gamma_t", None)
        dif = m.dynamics._forward(t, xh.clone().requires_grad_(True), batch["atom_mask"], em, None, None).detach()
        return inf, dif, tab, m.gamma(t).detach().view(-1, 1)
    inf0, dif0, tab0, net0 = both()
    assert rel_l2(inf0.cpu().numpy(), dif0.cpu().numpy()) < 2e-6 and torch.allclose(tab0, net0, rtol=1e-4, atol=1e-4)
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        m.training_step(batch, 0).backward()
        opt.step()
    inf1, dif1, tab1, net1 = both()
    assert rel_l2(dif1.cpu().numpy(), dif0.cpu().numpy()) > 1e-4, "the optimiser must have moved the weights"
    assert rel_l2(inf1.cpu().numpy(), dif1.cpu().numpy()) < 2e-6, "the inference handle runs on stale weights"
    moved = float((net1 - net0).abs().max())
    assert moved > 0 and float((tab1 - net1).abs().max()) < 0.25 * moved, "the schedule table is stale"


def test_inference_path_sees_writes_that_bump_no_version():
    """Round 6: a writer that is not a torch optimizer and bumps no version counter - `p.data.copy_`,
    `torch._foreach_add_` on `.data` - used to leave the packed weight image, the schedule table and the two-stream twin stale with
    no error (this test fails on the round-5 library).  Every key hit is now confirmed by a content digest of the parameters
    (csrc/k_digest.hpp, _lib.ImageGuard)."""
    from hierdiff_amd import DiffusionQM9, _lib, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, B, N = 64, 2, 5, 9
    m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L, timesteps=50))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 33, 0.5).items()})
    m = m.to(DEV).eval()
    batch = {k: v.to(DEV) for k, v in _host_batch(5, B, N).items()}
    xh = torch.randn(B, N, 11, device=DEV) * batch["atom_mask"]
    t = torch.full((B, 1), 0.5, device=DEV)
    em = batch["edge_mask"].reshape(B, N * N)

    def both():
        with torch.no_grad():
            inf = m.dynamics._forward(t, xh, batch["atom_mask"], em, None, None)
            tab = m._gamma_rows(t, "gamma_t", None)
        m.dynamics.differentiable = True
        dif = m.dynamics._forward(t, xh.clone().requires_grad_(True), batch["atom_mask"], em, None, None).detach()
        m.dynamics.differentiable = None
        return inf, dif, tab, m.gamma(t).detach().view(-1, 1)
    inf0, dif0, tab0, net0 = both()
    assert rel_l2(inf0.cpu().numpy(), dif0.cpu().numpy()) < 2e-6
    gen = _lib.optimizer_generation()
    versions = [p._version for p in m.parameters()]
    w = m.dynamics.egnn.e_block_1.gcl_0.edge_mlp[2].weight
    w.data.mul_(1.5)                                                   # .data: a separate version counter
    torch._foreach_add_([p.data for p in m.gamma.parameters()], 0.25)
    assert [p._version for p in m.parameters()] == versions and _lib.optimizer_generation() == gen, "the writes must bump nothing"
    inf1, dif1, tab1, net1 = both()
    assert rel_l2(dif1.cpu().numpy(), dif0.cpu().numpy()) > 1e-4, "the write must have moved the output"
    assert rel_l2(inf1.cpu().numpy(), dif1.cpu().numpy()) < 2e-6, "the inference handle runs on a stale weight image"
    moved = float((net1 - net0).abs().max())
    assert moved > 0 and float((tab1 - net1).abs().max()) < 0.25 * moved, "the schedule table is stale"
    # one ulp in one weight is seen; the digest does not depend on how the words are cut into tensors
    p0 = m.dynamics.egnn.embedding.weight
    d0 = _lib.params_digest([p0])
    flat = p0.detach().reshape(-1)
    assert _lib.params_digest([flat[:100], flat[100:]]) == d0 and _lib.params_digest([flat]) == d0
    assert _lib.params_digest([flat[:101], flat[101:103], flat[103:]]) == d0          # boundaries inside a group of four words
    bits = p0.data.view(torch.int32)
    bits[3, 2] += 1
    assert _lib.params_digest([p0]) != d0
    bits[3, 2] -= 1
    assert _lib.params_digest([p0]) == d0
    host = np.frombuffer(flat.cpu().numpy().tobytes(), dtype=np.uint32).astype(np.uint64)
    host = np.concatenate([host, np.zeros((-host.size) % 4, dtype=np.uint64)])
    pair = host[0::2] | (host[1::2] << np.uint64(32))
    x = pair + np.uint64(0x9E3779B97F4A7C15) * (np.arange(0, host.size, 2, dtype=np.uint64) + np.uint64(1))
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9); x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB); x ^= x >> np.uint64(31)
    want = int(np.sum(x, dtype=np.uint64))
    assert d0 == (want + 1) & 0xFFFFFFFFFFFFFFFF          # the kernel against a numpy restatement of csrc/k_digest.hpp


# ----------------------------------------------------------------------------- opt-in fp16x3 arithmetic of the training path

@pytest.mark.parametrize("H,rows", [(256, 4096), (256, 32 * 173), (128, 2048), (128, 32)])
def test_dw2_f16_matches_a_float64_product(H, rows):
    """k_dw2_f16 (round 5): dW2 = G2^T P on a two-way FP16 split of both operands, each ranged by ONE power of two taken from
    per-workgroup maxima.  The operands span what a backward pass produces - columns over four decades and ROWS over six (gradient
    rows of gated-off edges next to the ones that matter): an element far below its array's maximum loses bits in proportion to how
    little it contributes, so the reduction stays within 2e-6 of a float64 product; an all-zero operand, any
    number of maxima, deterministic."""
    from hierdiff_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(H + rows + 1)
    G2 = (torch.randn(rows, H, generator=g) * torch.logspace(-3, 1, H)[None, :] * torch.logspace(-6, 0, rows)[torch.randperm(rows, generator=g)][:, None]).to(DEV)
    P = (torch.randn(rows, H, generator=g) * 3.0).to(DEV)
    ref = G2.double().t() @ P.double()
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for slabs, nmax in ((256, max(1, rows // 128)), (7, 1), (1, 3)):
        # maxima per group of rows, as the backward stages leave them (any partition of the rows works: only the overall maximum matters)
        gm = torch.stack([c.abs().max() for c in G2.chunk(nmax)]).contiguous()
        pm = torch.stack([c.abs().max() for c in P.chunk(nmax)]).contiguous()
        dW2 = torch.full((H, H), float("nan"), device=DEV)
        ws = torch.empty(slabs * H * H, device=DEV)
        _lib.check(lib.hd_dw2_f16(0, rows, H, G2.data_ptr(), P.data_ptr(), gm.data_ptr(), pm.data_ptr(), gm.numel(), dW2.data_ptr(), H,
                                  ws.data_ptr(), ws.numel(), st), "hd_dw2_f16")
        err = float((dW2.double() - ref).norm() / ref.norm())
        f32 = float(((G2.t() @ P).double() - ref).norm() / ref.norm())
        print(f"hd_dw2_f16 H={H} rows={rows} slabs<={slabs} maxima {gm.numel()}: rel-L2 vs float64 {err:.2e} (torch fp32 matmul {f32:.2e})")
        assert err < 2e-6
        outs.append(dW2)
    again = torch.empty_like(outs[0])
    ws = torch.empty(256 * H * H, device=DEV)
    gm = torch.stack([c.abs().max() for c in G2.chunk(max(1, rows // 128))]).contiguous()
    pm = torch.stack([c.abs().max() for c in P.chunk(max(1, rows // 128))]).contiguous()
    _lib.check(lib.hd_dw2_f16(0, rows, H, G2.data_ptr(), P.data_ptr(), gm.data_ptr(), pm.data_ptr(), gm.numel(), again.data_ptr(), H,
                              ws.data_ptr(), ws.numel(), st), "hd_dw2_f16")
    assert torch.equal(again, outs[0])
    zero = torch.zeros_like(G2)
    zm = torch.zeros(1, device=DEV)
    _lib.check(lib.hd_dw2_f16(0, rows, H, zero.data_ptr(), P.data_ptr(), zm.data_ptr(), pm.data_ptr(), 1, again.data_ptr(), H,
                              ws.data_ptr(), ws.numel(), st), "hd_dw2_f16")
    assert float(again.abs().max()) == 0.0
    assert lib.hd_dw2_f16(0, 48, H, G2.data_ptr(), P.data_ptr(), gm.data_ptr(), pm.data_ptr(), 1, again.data_ptr(), H, ws.data_ptr(), ws.numel(), None) != 0
    assert lib.hd_dw2_f16(0, rows, H, G2.data_ptr(), P.data_ptr(), None, pm.data_ptr(), 1, again.data_ptr(), H, ws.data_ptr(), ws.numel(), None) != 0


@pytest.mark.parametrize("H", [256, 128])
def test_training_precision_fp16x3_gradients(H, monkeypatch):
    """`dynamics.training_precision = "fp16x3"` (round 5): forward contraction, stage B's dP = G2 W2 and dW2 = G2^T P in the two-way
    FP16 split (hd_edge_layer_forward_s / _backward_s with precision 3, hd_dw2_f16), ranges computed on the device from the data; the
    mode lives on top of the kept pre-activations, so the batch is one that keeps them (870 tiles) - and a 5-molecule batch falls
    back to exact fp32 layer by layer (round 6; bf16x6 before), with a one-time warning.  Bars: 1e-4 against the oracle's autograd,
    1e-5 against the exact-fp32 step (the fallback IS the exact-fp32 step: bit-equal).  Every parameter gradient and the input
    gradient - mixed precision without a loss of accuracy (the reference's own mixed mode is apex O2, conf/trainer/default.yaml:4-5)."""
    from hierdiff_amd import _lib
    from hierdiff_amd.weights import synthetic_state_dict
    L = 2
    lib = _lib.load()
    calls = {"fwd": [], "dw2": 0}
    of, od = lib.hd_edge_layer_forward_s, lib.hd_dw2_f16
    monkeypatch.setattr(lib, "hd_edge_layer_forward_s", lambda *a: (calls["fwd"].append(a[3]), of(*a))[1])
    monkeypatch.setattr(lib, "hd_dw2_f16", lambda *a: (calls.__setitem__("dw2", calls["dw2"] + 1), od(*a))[1])
    import warnings
    import hierdiff_amd.training as tr
    tr._WARNED.discard("fp16x3-fallback")
    for n_list, expect in (([30] * 30 + [17, 9], 3), ([30, 30, 17, 30, 9], 0)):
        sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 79, 0.5)
        cfg = orc.DynCfg(in_node_nf=9, hidden_nf=H, n_layers=L, normalization_factor=10.0)
        xh, nm, em = orc.random_inputs(n_list, 8, 74)
        B, N = xh.shape[:2]
        t = torch.linspace(0.1, 0.9, B).view(B, 1)
        w = torch.randn(B, N, 11, generator=torch.Generator().manual_seed(8))
        sd = _oracle_sd(sd_np)
        xo = xh.clone().requires_grad_(True)
        ref = orc.dynamics_forward(sd, cfg, t, xo, nm, em, None, None, prefix="dynamics.egnn.")
        (ref * w).sum().backward()
        grads = {}
        for mode in ("fp32", "fp16x3"):
            calls["fwd"].clear(); calls["dw2"] = 0
            dyn = build_dynamics(sd_np, H, L)
            dyn.precision = "fp32"
            dyn.training_precision = mode
            xg = xh.to(DEV).requires_grad_(True)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                out = dyn._forward(t.to(DEV), xg, nm.to(DEV), em.to(DEV), None, None)
            assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
            (out * w.to(DEV)).sum().backward()
            if mode == "fp16x3":
                assert calls["fwd"] and all(p == expect for p in calls["fwd"]), calls
                assert calls["dw2"] == (len(calls["fwd"]) if expect == 3 else 0)
                assert (expect == 0) == any("runs in exact fp32" in str(c.message) for c in caught)
            worst, n = _compare_grads(dyn.egnn.named_parameters(), sd, "dynamics.egnn.", f"H={H} training_precision={mode}")
            valid = nm.numpy()[..., 0]
            assert rel_l2(xg.grad.cpu().double().numpy()[valid], xo.grad.double().numpy()[valid]) < GRAD_TOL
            grads[mode] = {k: p.grad.detach().clone() for k, p in dyn.egnn.named_parameters()}
            print(f"H={H} B={B} training_precision={mode}: {n} tensors, worst grad rel-L2 vs oracle {worst:.2e}")
        between = max(float((grads["fp16x3"][k] - grads["fp32"][k]).norm() / grads["fp32"][k].norm().clamp_min(1e-30))
                      for k in grads["fp32"] if float(grads["fp32"][k].norm()) > 1e-6)
        print(f"H={H} B={B}: fp16x3 step (layers in precision {expect}) vs exact-fp32 step, worst parameter-gradient rel-L2 {between:.2e}")
        assert between < 1e-5 if expect == 3 else between == 0.0
    with pytest.raises(ValueError):
        dyn.training_precision = "bf16x3"


@pytest.mark.parametrize("mode", ["fp16x3"])
def test_training_precision_below_width_128_is_the_fp32_step(mode):
    """The split arithmetic exists from width 128 up; a narrower model asked for `training_precision = "fp16x3"` runs the
    exact-fp32 kernels (as `precision` does in sampling): same output and gradient bits as the fp32 step."""
    from hierdiff_amd.weights import synthetic_state_dict
    H, L = 64, 2
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 80, 0.5)
    xh, nm, em = orc.random_inputs([9, 4, 12, 7], 8, 75)
    B = xh.shape[0]
    t = torch.linspace(0.1, 0.9, B).view(B, 1)
    res = {}
    for m_ in ("fp32", mode):
        dyn = build_dynamics(sd_np, H, L)
        dyn.precision = "fp32"
        dyn.training_precision = m_
        xg = xh.to(DEV).requires_grad_(True)
        out = dyn._forward(t.to(DEV), xg, nm.to(DEV), em.to(DEV), None, None)
        out.square().sum().backward()
        res[m_] = (out.detach().clone(), xg.grad.clone(), [p.grad.clone() for p in dyn.egnn.parameters()])
    assert torch.equal(res["fp32"][0], res[mode][0]) and torch.equal(res["fp32"][1], res[mode][1])
    assert all(torch.equal(a, b) for a, b in zip(res["fp32"][2], res[mode][2]))


@pytest.mark.parametrize("H,B,mode", [(32, 5, "fp32"), (64, 7, "fp32"), (128, 32, "fp32"), (256, 32, "fp32"), (256, 6, "fp32")])
def test_kept_edge_activations_equal_the_recomputing_backward(H, B, mode, monkeypatch):
    """Round 5: the training forward keeps W2 P + b2 of every edge row (hd_edge_layer_forward_s) where the whole-tile edge kernel
    runs, and stage A of the backward pass loads it instead of recomputing it on the matrix cores (hd_edge_layer_backward_s).
    `dynamics.keep_edge_activations = False` is the recomputing path of rounds 2-4: same output bits, gradients equal to the
    bit in exact fp32 (the kept values ARE the recomputed ones: same arithmetic in the same order) - and the path under test is the
    one that ran: 870 tiles at B = 32 take the whole-tile kernel at widths 128 / 256, every batch does below 128, and the
    6-molecule batch at width 256 (column-split forward) keeps nothing."""
    from hierdiff_amd import _lib
    from hierdiff_amd.weights import synthetic_state_dict
    L = 2
    n_list = [30] * (B - 2) + [17, 9]
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 78, 0.5)
    xh, nm, em = orc.random_inputs(n_list, 8, 73)
    t = torch.linspace(0.1, 0.9, B).view(B, 1)
    w = torch.randn(B, xh.shape[1], 11, generator=torch.Generator().manual_seed(7))
    lib = _lib.load()
    orig = lib.hd_edge_layer_forward_s
    kept = []
    monkeypatch.setattr(lib, "hd_edge_layer_forward_s", lambda *a: (kept.append(a[13] is not None), orig(*a))[1])
    res = {}
    for keep in (True, False):
        kept.clear()
        dyn = build_dynamics(sd_np, H, L)
        dyn.precision = "fp32"
        dyn.training_precision = mode
        dyn.keep_edge_activations = keep
        xg = xh.to(DEV).requires_grad_(True)
        out = dyn._forward(t.to(DEV), xg, nm.to(DEV), em.to(DEV), None, None)
        (out * w.to(DEV)).sum().backward()
        expect = keep and not (H >= 128 and B < 32)
        assert kept and all(k == expect for k in kept), (keep, kept)
        res[keep] = (out.detach().clone(), xg.grad.clone(), {k: p.grad.detach().clone() for k, p in dyn.egnn.named_parameters()})
    assert torch.equal(res[True][0], res[False][0])
    worst = 0.0
    for k, g in res[False][2].items():
        d = float((res[True][2][k] - g).norm() / g.norm().clamp_min(1e-30))
        worst = max(worst, d)
    dx = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    print(f"H={H} B={B} {mode}: kept vs recomputed - worst parameter-gradient rel-L2 {worst:.2e}, d/dxh {dx:.2e}")
    assert worst == 0.0 and dx == 0.0


# ----------------------------------------------------------------------------- new masks every step: staged batches, pooled arenas
def _host_batch(seed, B=6, N=9, sizes=None):
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(3, N + 1, (B,), generator=g) if sizes is None else torch.as_tensor(sizes)
    nm = (torch.arange(N)[None, :] < sizes[:, None])[..., None]
    em = (nm.float() @ nm.float().transpose(1, 2)).bool() & ~torch.eye(N, dtype=torch.bool)[None]
    x = torch.randn(B, N, 3, generator=g) * nm
    x = x - (x.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
    h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2) * nm
    return {"positions": x, "atom_mask": nm, "edge_mask": em, "node_feature": h}


def _small_model(H=64, L=2, seed=31):
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    sd = synthetic_state_dict(9, 0, H, L, 2, True, seed, 0.5)
    m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m.to(DEV).train()


def test_stage_batch_builds_the_topology_from_the_host_masks(monkeypatch):
    """`DiffusionQM9.stage_batch(host batch)`: same loss and gradients, bit for bit, as moving the batch with `.to(device)` -
    and the step never copies a mask back to the host (the device-to-host copy of the fallback would stall a training loop
    behind everything the GPU has queued)."""
    import hierdiff_amd.dynamics as dyn_mod
    host = _host_batch(5)
    a, b = _small_model(), _small_model()
    torch.manual_seed(3)
    la = a.training_step({k: v.to(DEV) for k, v in host.items()}, 0)
    la.backward()
    calls = []
    real = dyn_mod.masks_to_host
    monkeypatch.setattr(dyn_mod, "masks_to_host", lambda *args: (calls.append(1), real(*args))[1])
    staged = b.stage_batch(host)
    assert all(v.device.type == "cuda" for v in staged.values())
    torch.manual_seed(3)
    lb = b.training_step(staged, 0)
    lb.backward()
    assert not calls, "the staged masks' topology must be found by identity"
    assert torch.equal(la.detach(), lb.detach())
    for (ka, pa), (kb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert (pa.grad is None) == (pb.grad is None) and (pa.grad is None or torch.equal(pa.grad, pb.grad)), ka
    # validation on staged masks takes the same shortcut
    b.eval()
    with torch.no_grad():
        torch.manual_seed(4)
        v1 = b.validation_step(b.stage_batch(host))["loss"]
        torch.manual_seed(4)
        v2 = b.validation_step({k: v.to(DEV) for k, v in host.items()})["loss"]
    assert torch.equal(v1, v2)
    with pytest.raises(Exception):
        b.dynamics.stage_masks(host["atom_mask"].to(DEV), host["edge_mask"].to(DEV))


def test_staging_many_small_tensors_never_waits_for_the_running_step():
    """tensors of one batch share a byte class of the pinned staging ring; the fifth of them used to WAIT for the
    first one's copy - queued behind the training step that is still running - i.e. for the GPU to drain.  A slot is now reused only
    when its copy has completed and the ring grows instead: a dozen stagings behind ~100 ms of queued GPU work return in
    milliseconds, and every copy carries its own bytes."""
    import time
    from hierdiff_amd.dynamics import _PIN_RING, _to_device_async
    dev = torch.device(DEV)
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(3):                      # library start-up and clock ramp outside the calibration
        a @ a
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(8):
        a @ a
    torch.cuda.synchronize(dev)
    per = (time.perf_counter() - t0) / 8
    reps = max(8, int(0.2 / per))
    srcs = [torch.full((257,), float(k)) + torch.arange(257) for k in range(12)]
    for _ in range(reps):
        a = (a @ a) * 1e-4
    t1 = time.perf_counter()
    outs = [_to_device_async(t, dev) for t in srcs]
    host = time.perf_counter() - t1
    torch.cuda.synchronize(dev)
    total = time.perf_counter() - t1
    assert total > 0.05, "the test needs GPU work in flight while it stages"
    assert host < 0.25 * total, f"staging waited for the GPU: {host * 1e3:.1f} ms of {total * 1e3:.1f} ms"
    for src, out in zip(srcs, outs):
        assert torch.equal(out.cpu(), src)
    ring = _PIN_RING[(4096, str(dev))]
    assert 4 <= len(ring[1]) <= 16384


def test_recycled_arenas_give_the_same_bits():
    """hd_topology_destroy hands a topology's arena to the pool, hd_topology_create_s takes the smallest that fits: a forward
    on a topology living in a RECYCLED arena (previous owner larger, different masks - stale tables and activations behind the
    new ones) equals the forward on a freshly allocated one bit for bit, on the launching stream and on a side stream."""
    import gc
    from hierdiff_amd import _lib
    from hierdiff_amd.weights import synthetic_state_dict
    H, L = 64, 2
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 77, 0.5)
    dyn = build_dynamics(sd_np, H, L)
    lib = _lib.load()

    def run(n_list, seed, stream=None):
        xh, nm, em = orc.random_inputs(n_list, 8, seed)
        B = xh.shape[0]
        t = torch.full((B, 1), 0.4)
        with torch.no_grad():
            if stream is None:
                return dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
            args = (t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV))
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                out = dyn._forward(*args, None, None)
            stream.synchronize()
            return out.cpu()

    def forget():
        dyn._topo_cache.clear(); dyn._topo_by_content.clear(); gc.collect()

    forget(); _lib.check(lib.hd_arena_pool_trim(), "trim")
    small = [7, 9, 4, 9]
    want = run(small, 11)                       # fresh arena
    forget(); _lib.check(lib.hd_arena_pool_trim(), "trim")
    run([30] * 12, 12)                          # a larger owner fills an arena with its tables and activations
    forget()                                    # ... and hands it to the pool
    got = run(small, 11)                        # recycled
    assert torch.equal(want, got)
    forget()
    side = torch.cuda.Stream()
    got_side = run(small, 11, side)             # recycled again, uploaded and used on a side stream
    assert torch.equal(want, got_side)
    # a topology created on one stream and used on another waits for its tables by itself
    forget()
    xh, nm, em = orc.random_inputs(small, 8, 11)
    args = (torch.full((4, 1), 0.4).to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV))
    torch.cuda.synchronize()
    with torch.no_grad():
        with torch.cuda.stream(side):
            dyn.topology(args[2], args[3], xh.shape[0], xh.shape[1])
        out = dyn._forward(*args, None, None)
    assert torch.equal(want, out.cpu())
    forget(); _lib.check(lib.hd_arena_pool_trim(), "trim")


def test_pipelined_fit_epoch_equals_step_by_step():
    """trainer.fit_epoch(..., device=...) stages batch k+1 (pinned copies + topology from the host masks) while step k runs:
    per-step losses, gradient norms and the final weights are bit-equal to ddp_step over `.to(device)` batches; 12 batches of
    never-repeated masks go through the 8-entry topology caches and the arena pool."""
    from hierdiff_amd.trainer import configure_optimizers, ddp_step, fit_epoch
    batches = [_host_batch(100 + k) for k in range(12)]
    a, b = _small_model(), _small_model()
    opt_a, _ = configure_optimizers(a)
    opt_b, _ = configure_optimizers(b)
    torch.manual_seed(9)
    log_a = fit_epoch(a, batches, opt_a, device=DEV)
    torch.manual_seed(9)
    log_b = [ddp_step(b, {k: v.to(DEV) for k, v in bt.items()}, opt_b) for bt in batches]
    assert log_a == log_b and len(log_a) == 12 and all(np.isfinite(r["loss"]) for r in log_a)
    for (ka, pa), (kb, pb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(pa, pb), ka
