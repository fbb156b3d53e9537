"""GPU tier (-m gpu): BASELINE.json's configurations at their stated shapes, the reference's edge cases, and the
round-2 golden fixtures (predefined schedules, 'elem' features, l2 loss, pocket loss) on the HIP path.

Every comparison goes through the C ABI (hierdiff_amd -> libhierdiff_hip.so) and checks against the CPU oracle or a
reference-generated golden vector; full-size cases add size-independent properties (bit-exact batch independence,
padding invariance, centre of gravity, finiteness).  Tolerance: the per-forward bar of tests/helpers.py (rel-L2 < 1e-4
over the whole output, max-abs < 1e-4 * max(1, max|ref|)) unless a test states otherwise.
"""
import copy
import os
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as orc
from tests.helpers import assert_parity, load, rel_l2
from tests.test_gpu_parity import DEV, PRECISIONS, build_diffusion, build_dynamics

pytestmark = pytest.mark.gpu

_CACHE = {}


def _syn(H, L, C_=0, seed=0, fin=9, pocket=False, gain=1.0):
    """Synthetic weights; `gain` scales the coordinate head (reference init 0.001, egnn_new.py:80-81).  gain 1.0 drives
    tanh(phi) into saturation (single-forward tests: exercises the tanh * coords_range path); trajectory tests use
    0.02, where the predicted velocity is O(1) like a trained model's - with gain 1.0 a T=20 chain runs away to
    |x| ~ 1e3 and amplifies any round-off difference chaotically."""
    from hierdiff_amd.weights import synthetic_state_dict
    return synthetic_state_dict(fin, C_, H, L, 2, True, seed, gain, pocket=pocket)


# ----------------------------------------------------------------------------- (a) the workload's length at production width

@pytest.mark.parametrize("precision", PRECISIONS)
def test_full_length_chain_production_width(precision):
    """T = 1000 posterior steps + decode at the production network (H=256, L=6, S=2) with injected normals, HIP path
    vs the CPU oracle (B=4, N=8: the oracle's 1001 forwards take ~15 s).  Bar on the final x and h: 1e-3 rel-L2 -
    the per-forward bar (1e-4) is not a trajectory bound: each step feeds its output error back through
    z_s = z_t/alpha - c*eps, and 1000 steps compound it; measured 2e-5 (fp32)."""
    from hierdiff_amd.noise_model import evaluate_gamma
    H, L, T = 256, 6, 1000
    n_list = [8, 5, 7, 3]
    sd_np = _syn(H, L, seed=21, gain=0.02)
    cfg = orc.DynCfg(hidden_nf=H, n_layers=L)
    nm, em = orc.canonical_masks(n_list)
    B, N = nm.shape[:2]
    g = torch.Generator().manual_seed(12)
    raws = [(torch.randn(B, N, 3, generator=g), torch.randn(B, N, 8, generator=g)) for _ in range(T + 2)]
    model = build_diffusion(sd_np, H, L, T=T, precision=precision)
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws)
    key = ("chain", H, L, T)
    if key not in _CACHE:      # the oracle replays the gamma table the product evaluates (fp64 on the host, rounded once)
        gg = evaluate_gamma(copy.deepcopy(model.gamma).cpu(), (torch.arange(T + 1, dtype=torch.float64) / T).view(-1, 1)).view(-1)
        with torch.no_grad():
            _CACHE[key] = orc.sample_chain(orc.as_torch_sd(sd_np), cfg, T, nm, em, None, raws, gamma_grid=gg)
    xo, ho = _CACHE[key]
    nmf = nm.float().numpy()
    rx = rel_l2(x.cpu().numpy() * nmf, xo.numpy() * nmf)
    rh = rel_l2(h.cpu().numpy(), ho.numpy())
    print(f"T=1000 H=256 L=6 chain [{precision}]: x {rx:.2e} h {rh:.2e}")
    assert rx < 1e-3 and rh < 1e-3
    assert torch.isfinite(x).all() and torch.isfinite(h).all()


# ----------------------------------------------------------------------------- (b) config 2: N=30, L=9 loop

@pytest.mark.parametrize("precision", PRECISIONS)
def test_config2_l9_loop_vs_oracle(precision):
    """BASELINE config 2 (full reverse diffusion, N=30 all valid, 9 EGNN layers) on a B=8, T=20 slice: the sampler
    loop incl. z_T, every posterior step and the decode against the oracle with the same injected normals."""
    from hierdiff_amd.noise_model import evaluate_gamma
    H, L, T, B, N = 256, 9, 20, 8, 30
    sd_np = _syn(H, L, seed=22, gain=0.02)
    cfg = orc.DynCfg(hidden_nf=H, n_layers=L)
    nm, em = orc.canonical_masks([N] * B)
    g = torch.Generator().manual_seed(13)
    raws = [(torch.randn(B, N, 3, generator=g), torch.randn(B, N, 8, generator=g)) for _ in range(T + 2)]
    model = build_diffusion(sd_np, H, L, T=T, precision=precision)
    x, h = model.sample_from_masks(nm.to(DEV), None, None, raw_noises=raws)
    key = ("cfg2", T)
    if key not in _CACHE:
        gg = evaluate_gamma(copy.deepcopy(model.gamma).cpu(), (torch.arange(T + 1, dtype=torch.float64) / T).view(-1, 1)).view(-1)
        with torch.no_grad():
            _CACHE[key] = orc.sample_chain(orc.as_torch_sd(sd_np), cfg, T, nm, em, None, raws, gamma_grid=gg)
    xo, ho = _CACHE[key]
    assert_parity(x.cpu().numpy(), xo.numpy(), f"config 2 x [{precision}]")
    assert_parity(h.cpu().numpy(), ho.numpy(), f"config 2 h [{precision}]")


@pytest.mark.parametrize("precision", PRECISIONS)
def test_config2_full_length_at_stated_size(precision):
    """BASELINE config 2 as stated: the FULL 1000-step reverse diffusion + decode, B = 64, N = 30 all valid, 9 EGNN layers,
    H = 256, library noise (the oracle would need hours for this; parity of the same loop is the T = 20 slice above and the
    T = 1000 chains at B = 4).  Size-independent properties of the complete run: every value finite, padded nothing, every
    molecule's centre of gravity at the origin, two runs bit-identical (no atomics anywhere), and - the batch cut in two
    shards with their global sample ids - every molecule bit-identical to the uncut run although the halves run other
    kernels (B = 64 in fp32: k_edge_mixed; B = 32: k_edge): what rank r of a sharded job computes."""
    H, L, T, B, N = 256, 9, 1000, 64, 30
    model = build_diffusion(_syn(H, L, seed=45, gain=0.02), H, L, T=T, precision=precision)
    nm = torch.ones(B, N, 1, dtype=torch.bool, device=DEV)
    x, h = model.sample_from_masks(nm, None, None, sample_id_base=900)
    assert torch.isfinite(x).all() and torch.isfinite(h).all()
    assert float(x.mean(1).abs().max()) < 1e-3 * max(1.0, float(x.abs().max()))
    x2, h2 = model.sample_from_masks(nm, None, None, sample_id_base=900)
    assert torch.equal(x, x2) and torch.equal(h, h2)
    for lo in (0, 32):
        xs, hs = model.sample_from_masks(nm[lo:lo + 32].contiguous(), None, None, sample_id_base=900 + lo)
        assert torch.equal(xs, x[lo:lo + 32]) and torch.equal(hs, h[lo:lo + 32]), f"shard at {lo}"
    print(f"config 2 full length [{precision}]: |x| max {float(x.abs().max()):.2f}, |h| max {float(h.abs().max()):.2f}")


@pytest.mark.parametrize("precision", PRECISIONS)
def test_config2_full_length_vs_oracle(precision):
    """BASELINE config 2 at its stated LENGTH against the oracle: the complete 1000-step reverse
    diffusion + decode at N = 30, 9 EGNN layers, H = 256 with injected normals.  A molecule's bits do not depend on its batch
    neighbours, so TWO molecules pin the chain: (i) the B = 64 run (k_edge_mixed + the small-batch node chain in fp32) and the
    same two molecules run alone (k_edge_split) are bit-identical, (ii) those two equal the CPU oracle's 1001-forward chain
    (~1 min on the host) within 1e-3 rel-L2 on the final x and h - the trajectory bar of test_full_length_chain_production_width."""
    from hierdiff_amd.noise_model import evaluate_gamma
    H, L, T, B, N = 256, 9, 1000, 64, 30
    sd_np = _syn(H, L, seed=46, gain=0.02)
    cfg = orc.DynCfg(hidden_nf=H, n_layers=L)
    g = torch.Generator().manual_seed(14)
    raws = [(torch.randn(B, N, 3, generator=g), torch.randn(B, N, 8, generator=g)) for _ in range(T + 2)]
    model = build_diffusion(sd_np, H, L, T=T, precision=precision)
    nm = torch.ones(B, N, 1, dtype=torch.bool, device=DEV)
    x, h = model.sample_from_masks(nm, None, None, raw_noises=raws)
    assert torch.isfinite(x).all() and torch.isfinite(h).all()
    pick = [5, 40]
    raws2 = [(rx[pick].contiguous(), rh[pick].contiguous()) for rx, rh in raws]
    x2, h2 = model.sample_from_masks(nm[:2].contiguous(), None, None, raw_noises=raws2)
    assert torch.equal(x2, x[pick]) and torch.equal(h2, h[pick])
    key = ("cfg2_full", T)
    if key not in _CACHE:
        nm2, em2 = orc.canonical_masks([N] * 2)
        gg = evaluate_gamma(copy.deepcopy(model.gamma).cpu(), (torch.arange(T + 1, dtype=torch.float64) / T).view(-1, 1)).view(-1)
        threads = torch.get_num_threads()
        torch.set_num_threads(min(8, threads))     # 60 x 30 x 30 edge rows per forward: a 128-thread pool is 10 x slower than 8 here
        try:
            with torch.no_grad():
                _CACHE[key] = orc.sample_chain(orc.as_torch_sd(sd_np), cfg, T, nm2, em2, None, raws2, gamma_grid=gg)
        finally:
            torch.set_num_threads(threads)
    xo, ho = _CACHE[key]
    rx = rel_l2(x2.cpu().numpy(), xo.numpy())
    rh = rel_l2(h2.cpu().numpy(), ho.numpy())
    print(f"config 2 full length vs oracle [{precision}]: x {rx:.2e} h {rh:.2e}")
    assert rx < 1e-3 and rh < 1e-3


# ----------------------------------------------------------------------------- (c) config 3: GEOM sizes padded to 48, B=256

def _geom_sizes(B, seed=2022, clip=48):
    from hierdiff_amd.geom_stats import GEOM_FRAGMENT_HISTOGRAM as HIST
    rng = np.random.Generator(np.random.PCG64(seed))
    keys = np.array([k for k in HIST if k <= clip])
    p = np.array([HIST[k] for k in keys], float)
    return [int(v) for v in rng.choice(keys, size=B, p=p / p.sum())]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_config3_geom_sizes_full_batch(precision):
    """BASELINE config 3 at its stated shape: B=256, n_b ~ conf/analyze/GEOM.yaml (seed 2022, clipped to 48), padded to
    N=48, L=6.  Full batch: finite, padded rows exactly 0, centre of gravity of the velocity ~ 0.  Parity: a slice of
    8 molecules (the largest, the smallest, 6 more) computed ALONE is bit-identical to its rows of the full batch
    (tiles are cut per molecule), and equals the oracle on that slice."""
    H, L, B, N = 256, 6, 256, 48
    n_list = _geom_sizes(B)
    assert max(n_list) <= 48 and min(n_list) >= 1
    sd_np = _syn(H, L, seed=23)
    xh, nm, em = orc.random_inputs(n_list, 8, 31, N)
    t = torch.linspace(0.02, 0.98, B).view(B, 1)
    dyn = build_dynamics(sd_np, H, L)
    dyn.precision = precision
    out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), None, None, None)
    assert torch.isfinite(out).all()
    o = out.cpu()
    assert np.all(o.numpy()[~nm.numpy()[..., 0]] == 0.0)
    assert (o[..., :3] * nm.float()).sum(1).abs().max().item() < 1e-4
    order = np.argsort(n_list)
    pick = sorted({int(order[-1]), int(order[0]), 3, 50, 97, 128, 200, 255})
    sub_n = [n_list[i] for i in pick]
    xs, nms, ems = xh[pick], nm[pick], em[pick]
    sub = dyn._forward(t[pick].to(DEV), xs.to(DEV), nms.to(DEV), None, None, None).cpu()
    assert torch.equal(sub, o[pick]), "a molecule's bits must not depend on its batch neighbours"
    with torch.no_grad():
        ref = orc.dynamics_forward(orc.as_torch_sd(sd_np), orc.DynCfg(hidden_nf=H, n_layers=L), t[pick], xs, nms, ems, None,
                                   None, prefix="dynamics.egnn.")
    assert_parity(sub.numpy(), ref.numpy(), f"config 3 slice n={sub_n} [{precision}]")
    # the same molecules padded to their own maximum instead of 48: identical bits again
    n2 = max(sub_n)
    tight = dyn._forward(t[pick].to(DEV), xs[:, :n2].contiguous().to(DEV), nms[:, :n2].contiguous().to(DEV), None, None, None).cpu()
    assert torch.equal(tight, sub[:, :n2])


# ----------------------------------------------------------------------------- (d) config 5 at its stated shape

@pytest.mark.parametrize("precision", PRECISIONS)
def test_config5_stated_shape_step_vs_oracle(precision):
    """BASELINE config 5: B=64, N=30, context feature, fix_noise, mol_shape=24 (last 6 nodes fixed, block-diagonal edge
    mask): one dynamics forward + one posterior step against the oracle (~3 s of CPU)."""
    H, L, B, N, mol = 256, 6, 64, 30, 24
    sd_np = _syn(H, L, C_=1, seed=24)
    cfg = orc.DynCfg(context_node_nf=1, hidden_nf=H, n_layers=L)
    rng = np.random.Generator(np.random.PCG64(77))
    node_mask = torch.ones(B, N, 1, dtype=torch.bool)
    edge_mask = torch.zeros(B, N, N, dtype=torch.bool)
    edge_mask[:, :mol, :mol] = True
    edge_mask[:, mol:, mol:] = True
    edge_mask &= ~torch.eye(N, dtype=torch.bool)[None]
    z = torch.from_numpy(rng.standard_normal((B, N, 11)).astype(np.float32))
    zx = orc.remove_mean_with_mask(z[:, :mol, :3], node_mask[:, :mol].float())
    z = torch.cat([torch.cat([zx, z[:, :mol, 3:]], dim=2), z[:, mol:]], dim=1)
    ctx = torch.zeros(B, N, 1) + torch.linspace(-0.4, 4.9, B).view(B, 1, 1)
    s = torch.full((B, 1), 299, dtype=torch.int64) / 1000
    t = torch.full((B, 1), 300, dtype=torch.int64) / 1000
    raw = (torch.from_numpy(rng.standard_normal((1, mol, 3)).astype(np.float32)),
           torch.from_numpy(rng.standard_normal((1, mol, 8)).astype(np.float32)))
    sd = orc.as_torch_sd(sd_np)
    model = build_diffusion(sd_np, H, L, C_=1, precision=precision)
    with torch.no_grad():
        gam = (orc.gamma_forward(sd, s), orc.gamma_forward(sd, t))
        ref_eps = orc.dynamics_forward(sd, cfg, t, z, node_mask, edge_mask, ctx, mol, prefix="dynamics.egnn.")
        ref_zs = orc.posterior_step(sd, cfg, s, t, z, node_mask, edge_mask, ctx, raw, mol_shape=mol, gammas=gam)
    eps = model.phi(z.to(DEV), t.to(DEV), node_mask.to(DEV), edge_mask.to(DEV), ctx.to(DEV), mol)
    assert_parity(eps.cpu().numpy(), ref_eps.numpy(), f"config 5 eps [{precision}]")
    zs = model.sample_p_zs_given_zt(s.to(DEV), t.to(DEV), z.to(DEV), node_mask.to(DEV), edge_mask.to(DEV), ctx.to(DEV),
                                    fix_noise=True, mol_shape=mol, raw_noise=raw, gammas=gam)
    assert tuple(zs.shape) == (B, mol, 11)
    assert_parity(zs.cpu().numpy(), ref_zs.numpy(), f"config 5 zs [{precision}]")
    # fixed rows of eps hold -mean(vel) (SURVEY.md appendix A quirk vi), like the reference
    assert ref_eps[:, mol:, :3].abs().max() > 0


# ----------------------------------------------------------------------------- (e) NaN guard

@pytest.mark.parametrize("precision", PRECISIONS)
def test_nan_guard_resets_whole_batch_velocity(precision):
    """en_dynamics.py:109-111: a NaN anywhere in the velocity zeroes the WHOLE batch's velocity (a warning is printed);
    the feature outputs keep their values - NaN only for the molecule that carried it.  The device-side guard
    (flag raised by k_post1, consumed by k_post2, no host sync) counts the event in hd_nan_events."""
    from hierdiff_amd import _lib
    H, L = 64, 2
    n_list = [9, 4, 7, 6, 3, 5]              # small molecules: several share a tail tile with the NaN molecule
    sd_np = _syn(H, L, seed=25)
    cfg = orc.DynCfg(hidden_nf=H, n_layers=L)
    xh, nm, em = orc.random_inputs(n_list, 8, 41)
    t = torch.full((len(n_list), 1), 0.3)
    for bad in (2, 0):                       # molecule 0 also feeds the padding rows of every tile
        xb = xh.clone()
        xb[bad, 1, 0] = float("nan")
        with torch.no_grad():
            ref = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xb, nm, em, None, None, prefix="dynamics.egnn.")
        assert torch.equal(ref[..., :3], torch.zeros_like(ref[..., :3]))
        dyn = build_dynamics(sd_np, H, L)
        dyn.precision = precision
        lib, cnt = _lib.load(), C.c_longlong()
        out = dyn._forward(t.to(DEV), xb.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
        _lib.check(lib.hd_nan_events(dyn._handle(), None, C.byref(cnt)))
        assert cnt.value == 1
        assert torch.equal(out[..., :3], torch.zeros_like(out[..., :3])), "velocity of the whole batch must be reset"
        # valid nodes: NaN exactly where the reference has it (every node of the poisoned molecule, nobody else).
        # Padded rows: the reference's dense edge list turns the padded rows of the poisoned molecule into NaN as well
        # (NaN * edge_mask 0); this implementation never computes padded nodes and keeps them exactly 0.
        valid = nm[..., 0]
        got_nan, ref_nan = torch.isnan(out[..., 3:]), torch.isnan(ref[..., 3:])
        assert torch.equal(got_nan[valid], ref_nan[valid]), "NaN must stay inside the molecule that carried it"
        assert ref_nan[bad][valid[bad]].all() and not ref_nan[[b for b in range(len(n_list)) if b != bad]].any()
        assert torch.equal(out[~valid], torch.zeros_like(out[~valid]))
        ok = valid.unsqueeze(-1) & ~ref_nan
        assert_parity(out[..., 3:][ok].numpy(), ref[..., 3:][ok].numpy(), f"NaN guard features [{precision}]")
        # the next (clean) forward is unaffected and does not count
        out2 = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
        _lib.check(lib.hd_nan_events(dyn._handle(), None, C.byref(cnt)))
        assert cnt.value == 1 and torch.isfinite(out2).all()
        with torch.no_grad():
            ref2 = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.")
        assert_parity(out2.numpy(), ref2.numpy(), "after NaN")
        # debug_checks prints the reference's warning (host sync)
        dyn.debug_checks = True
        import contextlib, io
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            dyn._forward(t.to(DEV), xb.to(DEV), nm.to(DEV), em.to(DEV), None, None)
        assert "detected nan" in buf.getvalue()


# ----------------------------------------------------------------------------- (f) the reference's runtime asserts

def test_debug_checks_reproduce_reference_asserts():
    """models/utils.py:47-50 (masked entries), :65-70 (centre of gravity), :73-75 (variables masked): opt-in
    (`debug_checks = True`, they cost host syncs); off by default."""
    H, L = 32, 1
    sd_np = _syn(H, L, seed=26)
    model = build_diffusion(sd_np, H, L, T=10)
    nm, em = orc.canonical_masks([5, 3, 4])
    xh, _, _ = orc.random_inputs([5, 3, 4], 8, 5)
    s = torch.full((3, 1), 4, dtype=torch.int64) / 10
    t = torch.full((3, 1), 5, dtype=torch.int64) / 10
    args = lambda z: (s.to(DEV), t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), None)
    model.sample_p_zs_given_zt(*args(xh))                              # clean input: fine either way
    shifted = xh.clone()
    shifted[..., 0] += 1.0 * nm[..., 0]                                 # centre of gravity off by 1
    model.sample_p_zs_given_zt(*args(shifted))                          # default: no host-side check
    model.debug_checks = True
    model.sample_p_zs_given_zt(*args(xh))
    with pytest.raises(AssertionError, match="Mean is not zero"):       # assert_mean_zero_with_mask, diffusion_qm9.py:328
        model.sample_p_zs_given_zt(*args(shifted))
    dirty = xh.clone()
    dirty[1, 4, 0] = 0.5                                                # node 4 of molecule 1 is padding
    with pytest.raises(AssertionError, match="not masked"):             # assert_correctly_masked inside it (:66)
        model.sample_p_zs_given_zt(*args(dirty))
    # DiffusionQM9.forward(batch): remove_mean_with_mask's own check (utils.py:47-50) and assert_correctly_masked (:740)
    x = xh[..., :3].clone()
    h = torch.cat([torch.randint(0, 5, (3, 5, 5)).float(), torch.randn(3, 5, 3)], dim=2) * nm
    batch = {"positions": x.to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.to(DEV), "node_feature": h.to(DEV)}
    torch.manual_seed(0)
    assert torch.isfinite(model(batch)["loss"])
    bad = dict(batch)
    bad["positions"] = dirty[..., :3].to(DEV)
    with pytest.raises(AssertionError, match="too high"):
        model(bad)
    model.debug_checks = False
    assert torch.isfinite(model(bad)["loss"])                           # unchecked, like a release build


# ----------------------------------------------------------------------------- (g) round-2 golden fixtures on the HIP path

def _raws(fx):
    return [(torch.from_numpy(fx["raw_x"][i]), torch.from_numpy(fx["raw_h"][i])) for i in range(len(fx["raw_x"]))]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_poly2_schedule_chain_and_l2_loss_golden(precision):
    """F12: PredefinedNoiseSchedule 'polynomial_2' + loss_type 'l2' (noise_model.py:125-160; diffusion_qm9.py:253-255,
    598-599, 611-612, 660-661): the reference's sample() chain and its training-mode loss value.  The schedule is a
    lookup table, so the product's own schedule path is exercised (no gamma injection)."""
    from hierdiff_amd import DiffusionQM9, default_config
    fx = load("f12_poly2_l2_h32_l2")
    H, L, T = int(fx["hidden_nf"]), int(fx["n_layers"]), int(fx["T"])
    cfg = default_config(hidden_nf=H, n_layers=L, timesteps=T)
    cfg["noise_schedule"] = "polynomial_2"
    cfg["loss_type"] = "l2"
    cfg["pre_noise"] = dict(noise_schedule="polynomial_2", timesteps=T, precision=1e-4)
    model = DiffusionQM9(cfg)
    sd_np = {k: v for k, v in _syn(H, L, seed=int(fx["weight_seed"])).items() if not k.startswith("gamma.")}
    sd_np["gamma.gamma"] = fx["gamma_table"]
    model.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()})
    model = model.to(DEV)
    model.dynamics.precision = precision
    n_list = [int(v) for v in fx["n_list"]]
    nm, em = orc.canonical_masks(n_list)
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=_raws(fx))
    assert_parity(x.cpu().numpy() * nm.float().numpy(), fx["x"], "poly2 chain x")
    assert_parity(h.cpu().numpy(), fx["h"], "poly2 chain h")
    model.train(True)
    loss, info = model.compute_loss(torch.from_numpy(fx["loss_x"]).to(DEV), torch.from_numpy(fx["loss_h"]).to(DEV), nm.to(DEV),
                                    em.to(DEV), None, t0_always=False, t_int=fx["t_int"], eps=fx["eps"])
    np.testing.assert_allclose(loss.cpu().numpy(), fx["loss"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(info["error"].cpu().numpy(), fx["error"], rtol=1e-4, atol=1e-5)
    assert float(fx["t_int"][0, 0]) == 0.0


@pytest.mark.parametrize("precision", PRECISIONS)
def test_elem_features_golden(precision):
    """F13: node_coarse_type 'elem' (3 node features, D = 6; diffusion_qm9.py:44-50, 470-476)."""
    from hierdiff_amd import DiffusionQM9, default_config
    fx = load("f13_elem_h64_l2")
    H, L, T = int(fx["hidden_nf"]), int(fx["n_layers"]), int(fx["T"])
    cfg = default_config(hidden_nf=H, n_layers=L, timesteps=T)
    cfg["node_coarse_type"] = "elem"
    model = DiffusionQM9(cfg)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in _syn(H, L, seed=int(fx["weight_seed"]), fin=4).items()})
    model = model.to(DEV)
    model.dynamics.precision = precision
    nm, em = torch.from_numpy(fx["node_mask"]), torch.from_numpy(fx["edge_mask"])
    out = model.phi(torch.from_numpy(fx["xh"]).to(DEV), torch.from_numpy(fx["t_rows"]).to(DEV), nm.to(DEV), em.to(DEV), None)
    assert tuple(out.shape)[-1] == 6
    assert_parity(out.cpu().numpy(), fx["out_row_t"], "elem forward")
    model.schedule_gammas = fx["gamma_grid"]
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=_raws(fx))
    assert_parity(x.cpu().numpy() * nm.float().numpy(), fx["x"], "elem chain x")
    assert_parity(h.cpu().numpy(), fx["h"], "elem chain h")
    model.T = 1000
    model.eval()
    gam = {k: fx[k] for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
    loss, _ = model.compute_loss(torch.from_numpy(fx["loss_x"]).to(DEV), torch.from_numpy(fx["loss_h"]).to(DEV), nm.to(DEV),
                                 em.to(DEV), None, t0_always=True, t_int=fx["t_int"], eps=fx["eps"], eps0=fx["eps0"], gammas=gam)
    np.testing.assert_allclose(loss.cpu().numpy(), fx["loss"], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_pocket_loss_golden(precision):
    """F14: DiffusionQM9.forward(batch) with cfg.pocket (diffusion_qm9.py:701-751 -> compute_loss with mol_shape < N)."""
    from hierdiff_amd import DiffusionQM9, default_config
    fx = load("f14_pocket_loss_h64_l2")
    H, L = int(fx["hidden_nf"]), int(fx["n_layers"])
    cfg = default_config(hidden_nf=H, n_layers=L)
    cfg["pocket"] = True
    model = DiffusionQM9(cfg)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in _syn(H, L, seed=int(fx["weight_seed"]), pocket=True).items()})
    model = model.to(DEV).eval()
    model.dynamics.precision = precision
    nm, em = orc.canonical_masks([int(v) for v in fx["n_list"]])
    batch = {"positions": torch.from_numpy(fx["positions"]).to(DEV), "atom_mask": nm.to(DEV), "edge_mask": em.to(DEV),
             "node_feature": torch.from_numpy(fx["node_feature"]).to(DEV), "protein_pos": torch.from_numpy(fx["pocket_pos"]).to(DEV),
             "protein_feat": torch.from_numpy(fx["pocket_feat"]).to(DEV), "protein_feat_mask": torch.from_numpy(fx["pocket_node_mask"]).to(DEV),
             "protein_edge_mask": torch.from_numpy(fx["pocket_edge_mask"]).to(DEV)}
    gam = {k: fx[k] for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
    out = model(batch, t_int=fx["t_int"], eps=fx["eps"], eps0=fx["eps0"], gammas=gam)
    assert abs(out["loss"].item() - float(fx["mean_loss"])) <= 1e-4 * abs(float(fx["mean_loss"])) + 1e-3


def test_switching_precision_keeps_the_schedule():
    """a new handle (other precision / device) must get the schedule again even if it re-uses the freed
    handle's address."""
    sd_np = _syn(128, 1, seed=27)                      # width 128: the fp16x3 mode runs its own node kernels (>= 128)
    model = build_diffusion(sd_np, 128, 1, T=6, precision="fp32")
    nm, _ = orc.canonical_masks([4, 3])
    x0, _ = model.sample_from_masks(nm.to(DEV), None, None)
    for p in ("fp16x3", "fp32", "fp16x3"):
        model.dynamics.precision = p
        x1, _ = model.sample_from_masks(nm.to(DEV), None, None)
        assert torch.isfinite(x1).all()
    model.dynamics.precision = "fp32"
    x2, _ = model.sample_from_masks(nm.to(DEV), None, None)
    assert torch.equal(x0, x2)


# ----------------------------------------------------------------------------- multi-process sharding on the GPU

def _shard_worker(rank, world, port, out_dir, n_list, T, base):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.sharding import broadcast_model_weights, shard_sample_ids
    torch.manual_seed(100 + rank)                        # ranks start from different random weights
    model = DiffusionQM9(default_config(hidden_nf=64, n_layers=2, timesteps=T))
    if rank == 0:
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in _syn(64, 2, seed=28).items()})
    broadcast_model_weights(model, src=0)                # the path's only collective (gloo here, RCCL under "nccl")
    model = model.to(DEV)
    lo, cnt = shard_sample_ids(0, len(n_list), rank, world)
    nm, _ = orc.canonical_masks(n_list[lo:lo + cnt])
    x, h = model.sample_from_masks(nm.to(DEV), None, None, sample_id_base=base + lo)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x.cpu().numpy(), h=h.cpu().numpy(), lo=lo, cnt=cnt)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_processes_reproduce_single_process_bits(tmp_path):
    """Two ranks (separate processes, one weight broadcast, contiguous global sample ids) against ONE process sampling
    the whole batch: every trajectory (T = 200 steps + decode) is bit-identical.  Both ranks share this box's single
    GPU and the broadcast runs over gloo; on an 8-GPU node the same code runs one rank per GPU over RCCL (bench.py)."""
    import socket
    import torch.multiprocessing as mp
    n_list, T, base = [7, 3, 9, 5, 1, 8, 12, 2, 6], 200, 4000
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path), n_list, T, base), nprocs=2, join=True)
    model = build_diffusion(_syn(64, 2, seed=28), 64, 2, T=T)
    nm, _ = orc.canonical_masks(n_list)
    x, h = model.sample_from_masks(nm.to(DEV), None, None, sample_id_base=base)
    x, h = x.cpu().numpy(), h.cpu().numpy()
    seen = 0
    for r in range(2):
        d = dict(np.load(tmp_path / f"rank{r}.npz"))
        lo, cnt, n2 = int(d["lo"]), int(d["cnt"]), d["x"].shape[1]
        assert np.array_equal(d["x"], x[lo:lo + cnt, :n2]) and np.array_equal(d["h"], h[lo:lo + cnt, :n2]), r
        seen += cnt
    assert seen == len(n_list)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_two_stream_sampler_reproduces_single_stream_bits(precision):
    """tests/two_stream.TwoStreamSampler (two half batches on two HIP streams, twin handle; a test harness since round 6) returns the bits of the
    plain sampler: ragged sizes, an odd batch, a context model, repeated calls (cached cuts / topologies / graphs) and a
    weight update in between."""
    from hierdiff_amd import EnVariationalDiffusion, default_config
    from tests.two_stream import TwoStreamSampler
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, T, N = 128, 2, 12, 12
    sd = synthetic_state_dict(9, 1, H, L, 2, True, 71, 1.0)
    m = EnVariationalDiffusion(default_config(hidden_nf=H, n_layers=L, context_node_nf=1, timesteps=T))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd.items()})
    m = m.to(DEV).eval()
    m.dynamics.precision = precision
    sizes = torch.tensor([12, 5, 9, 1, 12, 7, 3])
    nm = (torch.arange(N)[None, :] < sizes[:, None]).unsqueeze(-1).to(DEV)
    ctx = torch.linspace(-0.4, 4.9, 7).view(7, 1, 1).expand(7, N, 1).contiguous().to(DEV)
    two = TwoStreamSampler(m)
    with torch.no_grad():
        for rep in range(3):
            if rep == 2:
                m.dynamics.egnn.embedding.weight.mul_(1.01)          # the twin must follow the original's parameters
            x_ref, h_ref = m.sample_from_masks(nm, None, ctx, sample_id_base=40)
            x, h = two.sample_from_masks(nm, None, ctx, sample_id_base=40)
            torch.cuda.synchronize()
            assert torch.isfinite(x_ref).all() and torch.equal(x, x_ref) and torch.equal(h, h_ref), f"rep {rep}"


# ----------------------------------------------------------------------------- (k) config 4: B = 2048 as 8 shards, RCCL path

def test_config4_b2048_equals_eight_shards_of_256():
    """BASELINE.json config 4 at its stated size: ONE B = 2048 topology on one GPU against 8 shards of 256 molecules with
    sample_id_base = 256 r (what rank r of an 8-GPU job computes, hierdiff_amd/sharding.py), production network H = 256,
    L = 6, T = 20 posterior steps + decode, exact fp32: every molecule is bit-identical (`torch.equal`) - a sample's bits
    do not depend on the batch / rank / world size it is computed in.  N = 30 with ragged
    sizes so that tail tiles are shared between neighbouring molecules differently in the two layouts."""
    H, L, T, N, B, W = 256, 6, 20, 30, 2048, 8
    model = build_diffusion(_syn(H, L, seed=44, gain=0.02), H, L, T=T)
    g = torch.Generator().manual_seed(5)
    sizes = torch.randint(1, N + 1, (B,), generator=g)
    sizes[::3] = N                                        # a third of the batch at the full size
    nm = (torch.arange(N)[None, :] < sizes[:, None]).unsqueeze(-1).to(DEV)
    x_all, h_all = model.sample_from_masks(nm, None, None, sample_id_base=7000)
    assert torch.isfinite(x_all).all() and torch.isfinite(h_all).all()
    from hierdiff_amd.sharding import shard_sample_ids
    for r in range(W):
        lo, cnt = shard_sample_ids(7000, B, r, W)
        assert cnt == 256
        sl = slice(lo - 7000, lo - 7000 + cnt)
        x, h = model.sample_from_masks(nm[sl].contiguous(), None, None, sample_id_base=lo)
        assert torch.equal(x, x_all[sl]) and torch.equal(h, h_all[sl]), f"shard {r}"
    # centre of gravity of every molecule stays at the origin (size-independent property of the path)
    cog = (x_all * nm).sum(1) / sizes.view(-1, 1).to(DEV)
    assert float(cog.abs().max()) < 1e-3


def _run_bench(args, launcher, tmp_path, tag):
    """bench.py in a fresh process, plain or under torch.distributed.run with ONE rank; returns the parsed JSON line."""
    import json
    import socket
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    cmd = [sys.executable]
    if launcher:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [bench] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, f"{tag}: rc {p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"{tag}: expected ONE JSON line, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_rccl_path_at_world_one(tmp_path):
    """The N > 1 code path of bench.py, for real, on this box's single GPU: launched like the driver launches N = 8
    (python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1) the process initialises the "nccl" backend
    (= RCCL), broadcasts the packed weights with ncclBroadcast, brackets the timed region with RCCL barriers and
    all-gathers the per-rank times.  The line must report rccl_ranks and agree with a plain N = 1 run of the same
    workload within 5 % (same box, back to back; the 2 % the collectives could cost is inside the run-to-run spread)."""
    args = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--timesteps", "250", "--precision", "fp32", "--no-configs",
            "--no-cpu-baseline"]
    plain = _run_bench(args, False, tmp_path, "plain")
    rccl = _run_bench(args, True, tmp_path, "launcher")
    assert "rccl_ranks" not in plain
    assert rccl["rccl_ranks"] == 1 and rccl["rccl"]["backend"] == "nccl"
    assert rccl["rccl"]["broadcast_elements"] > 5_900_000            # 5,932,309 dynamics + 3,077 schedule parameters
    assert len(rccl["rank_elapsed_s"]) == 1 and rccl["rank_elapsed_s"][0] <= rccl["elapsed_max_s"] + 1e-6
    assert rccl["n_gpus"] == 1 and rccl["scaling"] == "weak"
    ratio = rccl["value"] / plain["value"]
    print(f"bench value plain {plain['value']} vs under the launcher with RCCL {rccl['value']} molecules/s (ratio {ratio:.3f})")
    assert 0.95 < ratio < 1.05
    for line in (plain, rccl):                                        # the contract's keys
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in line, k
        assert line["roofline"]["bound"] == "mfma" and 0.0 < line["roofline"]["frac"] < 1.0


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher environment starts its own ranks through torch.distributed.run
    (bench.self_launch).  On this box's one GPU the same path is taken with --force-launcher at N = 1: the child runs under the
    launcher (RCCL initialised, weights broadcast), the parent hands its JSON line and exit code through."""
    args = ["--gpus", "1", "--force-launcher", "--steps", "1", "--warmup", "1", "--timesteps", "100", "--precision", "fp32",
            "--no-configs", "--no-cpu-baseline"]
    line = _run_bench(args, False, tmp_path, "self-launch")
    assert line["rccl_ranks"] == 1 and line["rccl"]["backend"] == "nccl" and line["n_gpus"] == 1
    assert len(line["rank_elapsed_s"]) == 1 and line["value"] > 0


# ----------------------------------------------------------------------------- (l) the reference's shipped job: many small batches

@pytest.mark.parametrize("precision", ["fp32", "fp16x3"])
def test_merged_sample_batches_equal_the_loop(precision):
    """`sample_batches(batch_size=2, num_batches=16)` is the reference's shipped job (conf/sample/default.yaml:1-2,
    diffusion_qm9.py:397-436: 16 calls of sample(2)).  Here the 32 molecules run as one device batch (`merge_batches`);
    every molecule equals, bit for bit, what the reference's loop order produces (`merge_batches = 0`): sizes drawn batch
    by batch from the same generator state, global sample ids, a context value per batch.  Also with merge limits (molecules, edges) that
    cut the job into several device batches and with a width that takes the column-split small-batch kernels."""
    H, L, T = 128, 2, 15
    for ctx_nf, ctx_range in ((0, None), (1, [-0.4, 1.5, 4.9])):
        m = build_diffusion(_syn(H, L, C_=ctx_nf, seed=91, gain=0.02), H, L, C_=ctx_nf, T=T, precision=precision).eval()
        outs = []
        for merge, edges in ((0, 225_000), (256, 225_000), (6, 225_000), (2048, 700)):
            m.merge_batches, m.merge_edges = merge, edges
            torch.manual_seed(1234)                                   # the node-count draws of nodes_dist
            res, names = m.sample_batches(2, 16, DEV, context_range=ctx_range, sample_id_base=500)
            assert len(res) == 32 and names == []
            outs.append(res)
        assert len(outs) == 4
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert a["x"].shape == b["x"].shape and torch.isfinite(a["x"]).all()
                assert torch.equal(a["x"], b["x"]) and torch.equal(a["h"], b["h"])
                if ctx_nf:
                    assert torch.equal(a["context"], b["context"])
        if ctx_nf:                                                    # batch i carries context_range[i % 3] on every row
            for i, r in enumerate(outs[1]):
                assert float((r["context"] - ctx_range[(i // 2) % 3]).abs().max()) == 0.0


def test_sample_batches_with_mean_aggregation_runs_the_loop():
    """with `aggregation_method='mean'` the neighbour-sum divisor is the padded N of the CALL, so a molecule's
    result depends on the largest molecule of its batch: merging batches of different widths would change it.  `sample_batches`
    therefore runs the reference's loop for this configuration whatever `merge_batches` says - equal, bit for bit, to
    `merge_batches = 0` and to calling `sample()` batch by batch."""
    from hierdiff_amd import DiffusionQM9, default_config
    H, L, T = 64, 2, 8
    cfg = default_config(hidden_nf=H, n_layers=L, timesteps=T)
    cfg.dynamics.aggregation_method = "mean"
    m = DiffusionQM9(cfg)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in _syn(H, L, seed=92, gain=0.02).items()})
    m = m.to(DEV).eval()
    outs = []
    for merge in (4096, 0):
        m.merge_batches = merge
        torch.manual_seed(77)
        outs.append(m.sample_batches(3, 5, DEV, sample_id_base=40)[0])
    torch.manual_seed(77)
    by_hand = [r for i in range(5) for r in m.sample(3, DEV, sample_id_base=40 + 3 * i)]
    assert len({r["x"].shape[0] for r in by_hand}) > 1                # ragged sizes: the batches have different widths
    for a, b, c in zip(outs[0], outs[1], by_hand):
        assert torch.equal(a["x"], b["x"]) and torch.equal(a["h"], b["h"])
        assert torch.equal(a["x"], c["x"]) and torch.equal(a["h"], c["h"])


def test_randomised_parity_sweep():
    """The wide net next to the fixed cases: tests/fuzz_parity.py - random model shapes / options / batch compositions / edge masks
    for the forward (all three precision modes) and short sampling chains, against the CPU oracle (a 400 + 100 case run of the same
    script: profiles/r03_fuzz_parity.log)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "32", "5"], cwd=root,
                          capture_output=True, text=True, timeout=600)
    lines = proc.stdout.splitlines()
    tail = "\n".join(lines[-6:])
    assert proc.returncode == 0, tail + proc.stderr[-2000:]
    assert any("cases in" in ln and "failures 0," in ln for ln in lines), tail        # forward phase
    assert any("chains in" in ln and "failures so far 0" in ln for ln in lines), tail   # chains
    assert any("batch splits in" in ln and "failures so far 0" in ln for ln in lines), tail   # batch splits / padding
    assert "failures in total 0" in lines[-1], tail                                    # gnn_dynamics


def test_mfma_probe_reports_a_plausible_sustained_rate():
    """hd_mfma_probe (bench.py's roofline.sustained): ns per matrix instruction and SIMD with every SIMD streaming it from registers.
    One instruction occupies the pipe for 64 (fp32) or 32 (fp16) cycles, so at 1.2 - 2.5 GHz the figure lies in 25 - 55 ns
    resp. 12 - 30 ns; bad arguments are refused with a message."""
    import ctypes as C
    from hierdiff_amd import _lib
    lib = _lib.load()
    data = (torch.rand(1024) * 2 - 1).to(DEV)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    scratch = torch.empty(2 * 256 * n_cu, device=DEV)
    ns = C.c_double()
    stream = torch.cuda.current_stream().cuda_stream
    for kind, lo, hi in ((0, 25.0, 55.0), (1, 12.0, 30.0)):
        _lib.check(lib.hd_mfma_probe(0, kind, data.data_ptr(), scratch.data_ptr(), 2000, C.byref(ns), stream), "hd_mfma_probe")
        assert lo < ns.value < hi, (kind, ns.value)
    for bad in (7, 2):            # (2 was the bf16 instruction up to ABI 11)
        assert lib.hd_mfma_probe(0, bad, data.data_ptr(), scratch.data_ptr(), 100, C.byref(ns), stream) != 0
    assert b"kind" in lib.hd_last_error()
    assert lib.hd_mfma_probe(0, 1, None, scratch.data_ptr(), 100, C.byref(ns), stream) != 0
