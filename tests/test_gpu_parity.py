"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the golden vectors generated from
the reference and against the CPU oracle on seeded inputs.

Tolerance: rel-L2 over the whole [B,N,3+F] output < 1e-4 and
max-abs < 1e-4*max(1, max|ref|); masked rows bit-exact 0.  A correct fp32 kernel lands near 1e-6.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as orc
from tests.helpers import assert_parity, fixture_model, load, rel_l2

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def build_dynamics(sd_np, H, L, C_=0, **kw):
    from hierdiff_amd import EGNN_dynamics_QM9
    args = dict(hidden_nf=H, n_layers=L, attention=True, tanh=True, normalization_factor=10, inv_sublayers=2)
    args.update(kw)
    m = EGNN_dynamics_QM9(9, C_, 3, **args)
    m.load_numpy_state_dict(sd_np, prefix="dynamics.")
    return m.to(DEV)


def build_diffusion(sd_np, H, L, C_=0, T=1000, precision=None):
    from hierdiff_amd import DiffusionQM9, default_config
    m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L, context_node_nf=C_, timesteps=T))
    m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()})
    if precision is not None:
        m.dynamics.precision = precision
    return m.to(DEV)


def test_library_loaded_and_gpu_visible():
    from hierdiff_amd import _lib
    _lib.require_gpu()
    assert torch.cuda.is_available()


FORWARD_FIXTURES = ["f1_cfg1_h256_l3", "f1b_cfg1_h256_l3_gain1", "f7_h32_l2", "f7_h64_l2", "f7_h128_l1",
                    "f6_b16_n30_h256_l9", "f6b_n48_h256_l6"]


PRECISIONS = ["fp32", "fp16x3"]     # exact-fp32 matrix path / two-way FP16 split (three MFMAs per product); one bar


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", FORWARD_FIXTURES)
def test_forward_golden(name, precision):
    fx = load(name)
    sd_np, _, _ = fixture_model(fx)
    dyn = build_dynamics(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]))
    dyn.precision = precision
    xh = torch.from_numpy(fx["xh"]).to(DEV)
    nm = torch.from_numpy(fx["node_mask"]).to(DEV)
    em = torch.from_numpy(fx["edge_mask"]).to(DEV)
    B, N = xh.shape[:2]
    worst = 0.0
    for k, tv in enumerate(fx["t_values"]):
        out = dyn._forward(torch.full((B, 1), float(tv), device=DEV), xh, nm, em, None, None)
        r, _ = assert_parity(out.cpu().numpy(), fx[f"out_t{k}"], f"{name} t={tv}")
        worst = max(worst, r)
    out = dyn._forward(torch.tensor([float(fx["t_values"][0])], device=DEV), xh, nm, em, None, N)
    assert_parity(out.cpu().numpy(), fx["out_scalar_t"], f"{name} scalar t")
    out = dyn._forward(torch.from_numpy(fx["t_rows"]).to(DEV), xh, nm, em, None, None)
    assert_parity(out.cpu().numpy(), fx["out_row_t"], f"{name} row t")
    o = out.cpu().numpy()
    assert np.all(o[~fx["node_mask"][..., 0]] == 0.0), "masked rows must be exactly zero"
    # canonical masks: passing edge_mask=None-equivalent topology gives the same bits
    topo = dyn.topology(nm, None, B, N)
    out2 = dyn.forward_with_topology(topo, torch.from_numpy(fx["t_rows"]).to(DEV), xh, None, None)
    assert torch.equal(out, out2)
    print(f"{name} [{precision}]: worst rel_l2 {worst:.2e}")


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["f19_mean_h64_l2", "f19_mean_h256_l3"])
def test_mean_aggregation_golden(name, precision):
    """F19: aggregation_method='mean' (egnn_new.py:283-288) - neighbour sums divided by the edge-list entries per node (the padded
    N) instead of normalization_factor - on canonical / general edge masks and with fixed nodes, in every precision mode."""
    fx = load(name)
    sd_np, _, _ = fixture_model(fx)
    dyn = build_dynamics(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), aggregation_method="mean")
    dyn.precision = precision
    xh, nm, t = (torch.from_numpy(fx[k]).to(DEV) for k in ("xh", "node_mask", "t_rows"))
    for tag, mask, mol in (("canonical", "edge_mask", None), ("general", "edge_mask_general", None),
                           ("fixed_nodes", "edge_mask", int(fx["mol_shape_fixed"]))):
        out = dyn._forward(t, xh, nm, torch.from_numpy(fx[mask]).to(DEV), None, mol)
        assert_parity(out.cpu().numpy(), fx["out_" + tag], f"{name} {tag}")
    # 'mean' on the reference's all-pairs edge list is the 'sum' arithmetic with normalization_factor = padded N: same bits
    dyn_n = build_dynamics(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), normalization_factor=xh.shape[1])
    dyn_n.precision = precision
    em = torch.from_numpy(fx["edge_mask"]).to(DEV)
    assert torch.equal(dyn._forward(t, xh, nm, em, None, None), dyn_n._forward(t, xh, nm, em, None, None))
    dyn_s = build_dynamics(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]))          # 'sum' / 10 must not reproduce it
    assert rel_l2(dyn_s._forward(t, xh, nm, em, None, None).cpu().numpy(), fx["out_canonical"]) > 1e-3


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["f3_cond_h256_l3", "f3_cond_h32_l2"])
def test_conditional_step_golden(name, precision):
    fx = load(name)
    sd_np, _, _ = fixture_model(fx, context_node_nf=1)
    model = build_diffusion(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), C_=1, precision=precision)
    z, nm, em, ctx = (torch.from_numpy(fx[k]).to(DEV) for k in ("z", "node_mask", "edge_mask", "context"))
    mol = int(fx["mol_shape"])
    s, t = torch.from_numpy(fx["s"]).to(DEV), torch.from_numpy(fx["t"]).to(DEV)
    eps = model.phi(z, t, nm, em, ctx, mol)
    assert_parity(eps.cpu().numpy(), fx["eps"], name + " eps")
    zs = model.sample_p_zs_given_zt(s, t, z, nm, em, ctx, fix_noise=True, mol_shape=mol,
                                    raw_noise=(torch.from_numpy(fx["raw_x"]), torch.from_numpy(fx["raw_h"])),
                                    gammas=(torch.from_numpy(fx["gamma_s"]), torch.from_numpy(fx["gamma_t"])))
    assert tuple(zs.shape) == tuple(fx["zs"].shape)
    assert_parity(zs.cpu().numpy(), fx["zs"], name + " zs")


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name,graph", [("f5_chain_h256_l3", False), ("f5_chain_h256_l3", True),
                                        ("f5_chain_h32_l2", True)])
def test_chain_golden(name, graph, precision):
    fx = load(name)
    sd_np, _, _ = fixture_model(fx)
    T = int(fx["T"])
    model = build_diffusion(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), T=T, precision=precision)
    model.use_graph = graph
    model.schedule_gammas = fx["gamma_grid"]        # replay the schedule values the reference run used
    n_list = [int(v) for v in fx["n_list"]]
    nm, em = orc.canonical_masks(n_list)
    raws = [(torch.from_numpy(fx["raw_x"][i]), torch.from_numpy(fx["raw_h"][i])) for i in range(T + 2)]
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws)
    nmf = nm.float().numpy()
    assert_parity(x.cpu().numpy() * nmf, fx["x"], name + " x")
    assert_parity(h.cpu().numpy(), fx["h"], name + " h")


@pytest.mark.parametrize("schedule", ["recorded", "fp64", "fp32"])
def test_long_chain_golden_and_default_schedule(schedule):
    """F16: the reference's own T = 1000 chain (H=32, L=2, exact fp32).  "recorded": with the reference's gamma grid
    injected the HIP trajectory ends within 1e-3 of the reference's x / h (the kernels).  "fp64": the product's DEFAULT
    schedule path, nothing injected - GammaNetwork evaluated once in float64 on the host - ends within the stated bound 1e-2
    (measured on the oracle: 3.6e-3 x / 2.7e-3 h, tests/test_oracle_golden.py): that is the end-to-end effect of the
    schedule deviation.  "fp32": the opt-in reference-style evaluation (`schedule_eval = "fp32"`, host BLAS dependent) -
    same bound on a foreign host, run-for-run agreement on the reference's own."""
    from tests.helpers import chain_noise, rel_l2
    fx = load("f16_chain_T1000_h32_l2")
    sd_np, _, _ = fixture_model(fx)
    T = int(fx["T"])
    model = build_diffusion(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), T=T, precision="fp32")
    if schedule == "recorded":
        model.schedule_gammas = fx["gamma_grid"]
    else:
        model.schedule_eval = schedule
    n_list = [int(v) for v in fx["n_list"]]
    nm, em = orc.canonical_masks(n_list)
    raws = chain_noise(fx["noise_seed"], T, len(n_list), max(n_list))
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws)
    nmf = nm.float().numpy()
    rx, rh = rel_l2(x.cpu().numpy() * nmf, fx["x"]), rel_l2(h.cpu().numpy(), fx["h"])
    print(f"F16 T=1000 chain, schedule {schedule}: x {rx:.2e} h {rh:.2e}")
    bound = 1e-3 if schedule == "recorded" else 1e-2
    assert rx < bound and rh < bound


def _oracle_case(n_list, H, L, seed, n_max=None, coord_gain=1.0, C_=0):
    from hierdiff_amd.weights import synthetic_state_dict
    sd_np = synthetic_state_dict(9, C_, H, L, 2, True, seed, coord_gain)
    cfg = orc.DynCfg(in_node_nf=9, context_node_nf=C_, hidden_nf=H, n_layers=L, normalization_factor=10.0)
    xh, nm, em = orc.random_inputs(n_list, 8, seed + 1, n_max)
    return sd_np, orc.as_torch_sd(sd_np), cfg, xh, nm, em


@pytest.mark.parametrize("n_list,H,L,n_max", [
    ([1], 32, 1, None),                       # single node, zero edges
    ([2, 2, 2, 1, 2], 32, 2, 3),              # one-edge segments, many segments per tile
    ([33, 40, 5], 64, 2, 40),                 # segments spanning 2-3 tiles
    ([30] * 8, 256, 2, None),                 # headline shape slice
    ([83, 3], 128, 1, None),                  # largest GEOM molecule
    ([17, 9, 30, 12, 25, 7, 14, 21], 256, 6, 48),   # config-3 flavour, production depth
])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_forward_vs_oracle(n_list, H, L, n_max, precision):
    sd_np, sd, cfg, xh, nm, em = _oracle_case(n_list, H, L, seed=40 + len(n_list), n_max=n_max)
    B, N = xh.shape[:2]
    t = torch.linspace(0.1, 0.9, B).view(B, 1)
    with torch.no_grad():
        ref = orc.dynamics_forward(sd, cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.")
    dyn = build_dynamics(sd_np, H, L)
    dyn.precision = precision
    out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
    assert_parity(out.numpy(), ref.numpy(), f"n={n_list} H={H} L={L} {precision}")
    assert np.all(out.numpy()[~nm.numpy()[..., 0]] == 0.0)
    vel_sum = (out[:, :, :3] * nm.float()).sum(1).abs().max().item()
    bound = 2e-6 * N * max(1.0, out[:, :, :3].abs().max().item())
    assert vel_sum < bound, f"centre of gravity of vel {vel_sum} (bound {bound})"


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("scale", [30.0, 300.0])
def test_saturating_activations_vs_oracle(scale, precision):
    """Coordinates far from the origin push the edge pre-activations beyond +-88 (exp overflow range):
    SiLU must saturate to 0 / x like the reference instead of producing NaN."""
    sd_np, sd, cfg, xh, nm, em = _oracle_case([12, 20, 7], 64, 2, seed=91)
    xh = torch.cat([xh[..., :3] * scale, xh[..., 3:]], dim=-1)
    t = torch.full((3, 1), 0.05)
    with torch.no_grad():
        ref = orc.dynamics_forward(sd, cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.")
    assert torch.isfinite(ref).all()
    dyn = build_dynamics(sd_np, 64, 2)
    dyn.precision = precision
    out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
    assert_parity(out.numpy(), ref.numpy(), f"saturating scale={scale} {precision}")


def test_fp16x3_is_fp32_accurate():
    """The two-way FP16 split is offered as an fp32-ACCURATE mode (the fp32 MFMA is not a matrix-core instruction on gfx950,
    DESIGN.md section 4b): its distance to the float64 evaluation of the oracle must be that of the exact-fp32 mode and of
    the float32 reference itself (all three are dominated by fp32 accumulation, ~2e-7), on the headline width."""
    sd_np, sd, cfg, xh, nm, em = _oracle_case([30] * 6 + [17, 9], 256, 3, seed=404)
    t = torch.full((8, 1), 0.3)
    with torch.no_grad():
        ref32 = orc.dynamics_forward(sd, cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.").numpy()
        with orc.float64():
            ref64 = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.").numpy()
    err = {}
    for precision in PRECISIONS:
        dyn = build_dynamics(sd_np, 256, 3)
        dyn.precision = precision
        out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu().numpy()
        err[precision] = rel_l2(out, ref64)
    err["float32 reference"] = rel_l2(ref32, ref64)
    print("distance to the float64 oracle:", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["fp16x3"] < 1.5 * max(err["fp32"], err["float32 reference"]), err
    assert err["fp32"] < 2e-6 and err["fp16x3"] < 2e-6


def test_retired_precisions_are_rejected_with_a_pointer():
    """Round 6: "bf16x3" / "bf16x6" no longer exist (ABI 12); the Python mirror says what to use instead, and the library rejects
    the old codes at hd_create."""
    import ctypes as C
    from hierdiff_amd import _lib
    sd_np, _, _, _, _, _ = _oracle_case([5, 3], 32, 1, seed=1)
    dyn = build_dynamics(sd_np, 32, 1)
    for name in ("bf16x3", "bf16x6"):
        with pytest.raises(ValueError, match="fp16x3"):
            dyn.precision = name
    with pytest.raises(ValueError):
        dyn.training_precision = "bf16x6"
    lib = _lib.load()
    for code in (1, 2):
        cfg = _lib.HdConfig(9, 0, 3, 32, 1, 2, 1, 1, 1, 0.0, 10.0, 30.0, code, 0)
        h = C.c_void_p()
        assert lib.hd_create(C.byref(cfg), 0, C.byref(h)) == -1 and b"retired" in lib.hd_last_error()


@pytest.mark.parametrize("w2_gain,first_gain", [(2.0 ** -9 * 0.7, 1.0), (2.0 ** 7 * 1.3, 1.0), (1.0, 2.0 ** -8), (1.0, 40.0)])
def test_fp16x3_operand_ranging(w2_gain, first_gain):
    """fp16 has five exponent bits; the mode brings its operands into range by exact powers of two - the W2 image per matrix,
    the activations per edge row from a bound on the pre-activation (k_ab_rowmax, k_edge.hpp).  Second-layer weights 700 x
    smaller / 170 x larger than usual, first-layer terms 256 x smaller / 40 x larger: the distance to the float64 oracle stays
    that of the exact-fp32 mode (a fixed scaling would lose 2 - 3 digits at either end: scratch/mb/f16_denorm.hip)."""
    sd_np, sd, cfg, xh, nm, em = _oracle_case([30, 30, 17, 9], 256, 2, seed=505)
    for k in list(sd_np):
        if k.endswith("edge_mlp.2.weight") or k.endswith("coord_mlp.2.weight"):
            sd_np[k] = (sd_np[k] * w2_gain).astype(np.float32)
        if k.endswith("edge_mlp.0.weight") or k.endswith("edge_mlp.0.bias") or k.endswith("coord_mlp.0.weight") or k.endswith("coord_mlp.0.bias"):
            sd_np[k] = (sd_np[k] * first_gain).astype(np.float32)
    t = torch.full((4, 1), 0.4)
    with torch.no_grad(), orc.float64():
        ref64 = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.").numpy()
    err = {}
    for precision in ("fp32", "fp16x3"):
        dyn = build_dynamics(sd_np, 256, 2)
        dyn.precision = precision
        out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu().numpy()
        assert np.isfinite(out).all()
        err[precision] = rel_l2(out, ref64)
    print(f"W2 x {w2_gain:.3g}, first layer x {first_gain:.3g}: distance to the float64 oracle", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["fp16x3"] < max(2.0 * err["fp32"], 1e-6), err


@pytest.mark.parametrize("node_gain,emb_gain", [(2.0 ** -8, 1.0), (60.0, 1.0), (1.0, 2.0 ** -10), (1.0, 3000.0)])
def test_fp16x3_node_operand_ranging(node_gain, emb_gain):
    """The node update of the mode (k_node<..., F16>, width 256) ranges its three activation operands per row from a-priori bounds
    (row maximum of [h | agg], L1 norms of W3 / W4) and its weights per matrix: node-MLP weights 256 x smaller / 60 x larger,
    node features 1000 x smaller / 3000 x larger (|h| up to ~1e5, beyond FP16's 65504) stay at the exact-fp32 mode's distance
    to the float64 oracle."""
    sd_np, sd, cfg, xh, nm, em = _oracle_case([30, 30, 17, 9], 256, 2, seed=707)
    for k in list(sd_np):
        if ".node_mlp." in k:
            sd_np[k] = (sd_np[k] * node_gain).astype(np.float32)
        if k.endswith("egnn.embedding.weight") or k.endswith("egnn.embedding.bias"):
            sd_np[k] = (sd_np[k] * emb_gain).astype(np.float32)
    t = torch.full((4, 1), 0.4)
    with torch.no_grad(), orc.float64():
        ref64 = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.").numpy()
    err = {}
    for precision in ("fp32", "fp16x3"):
        dyn = build_dynamics(sd_np, 256, 2)
        dyn.precision = precision
        out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu().numpy()
        assert np.isfinite(out).all()
        err[precision] = rel_l2(out, ref64)
    print(f"node MLP x {node_gain:.3g}, embedding x {emb_gain:.3g}: distance to the float64 oracle", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["fp16x3"] < max(2.0 * err["fp32"], 1e-6), err


def test_fp16x3_has_no_range_limit():
    """FP16 overflows at 65504; the mode's activations are ranged per edge row by max|A_i| + max|B_j| + the distance terms
    (k_ab_rowmax + k_edge.hpp), so first-layer terms far beyond that - a bias of 60000, coordinates 1000 apart - are computed
    like in exact fp32: no inf, no NaN event, same distance to the float64 oracle."""
    from hierdiff_amd import _lib
    sd_np, sd, cfg, xh, nm, em = _oracle_case([9, 12], 128, 1, seed=606)
    for k in list(sd_np):
        if k.endswith("edge_mlp.0.bias"):
            sd_np[k] = (sd_np[k] + 60000.0).astype(np.float32)
    xh = torch.cat([xh[..., :3] * 300.0, xh[..., 3:]], dim=-1)
    t = torch.full((2, 1), 0.4)
    with torch.no_grad(), orc.float64():
        ref64 = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.").numpy()
    err = {}
    for precision in ("fp32", "fp16x3"):
        dyn = build_dynamics(sd_np, 128, 1)
        dyn.precision = precision
        out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
        cnt = C.c_longlong()
        _lib.check(_lib.load().hd_nan_events(dyn._handle(), torch.cuda.current_stream().cuda_stream, C.byref(cnt)))
        assert cnt.value == 0 and torch.isfinite(out).all() and float(out.abs().max()) > 0
        err[precision] = rel_l2(out.numpy(), ref64)
    print("bias 60000, coordinates x 300: distance to the float64 oracle", {k: f"{v:.2e}" for k, v in err.items()})
    assert err["fp16x3"] < max(2.0 * err["fp32"], 1e-6), err


def test_general_edge_mask_and_options_vs_oracle():
    """Arbitrary edge_mask (block-diagonal with self edges and a masked-out valid pair), attention off,
    tanh off, norm_constant != 0, three sublayers."""
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, S = 64, 2, 3
    sd_np = synthetic_state_dict(9, 0, H, L, S, False, 77, 1.0)
    cfg = orc.DynCfg(in_node_nf=9, hidden_nf=H, n_layers=L, inv_sublayers=S, attention=False, tanh=False,
                     norm_constant=1.0, normalization_factor=3.0)
    xh, nm, em = orc.random_inputs([9, 6, 12], 8, 5, 12)
    em = em.clone()
    em[0, :4, 4:9] = False; em[0, 4:9, :4] = False     # two disconnected blocks
    em[1, 2, 2] = True                                  # a self edge
    em[2, 0, 1] = False                                 # asymmetric hole
    t = torch.tensor([[0.3], [0.6], [0.9]])
    with torch.no_grad():
        ref = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.")
    from hierdiff_amd import EGNN_dynamics_QM9
    dyn = EGNN_dynamics_QM9(9, 0, 3, hidden_nf=H, n_layers=L, attention=False, tanh=False, norm_constant=1.0,
                            inv_sublayers=S, normalization_factor=3.0)
    dyn.load_numpy_state_dict(sd_np, prefix="dynamics.")
    dyn = dyn.to(DEV)
    out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
    assert_parity(out.numpy(), ref.numpy(), "general edge mask")


@pytest.mark.parametrize("precision,H", [("fp32", 64), ("fp16x3", 64), ("fp16x3", 128)])
def test_pocket_sized_graph_vs_oracle(precision, H):
    """Pocket-conditioned jobs put the ligand fragments AND the pocket residues into one graph (diffusion_qm9.py:362-371):
    N in the hundreds, a node's edges span 7 tiles, ligand rows fixed through mol_shape.  N = 200 / 137, dense edges plus
    a block the mask removes, against the oracle.  fp16x3 at both of its node paths (width 64: fp32 node kernels + k_ab_rowmax;
    width 128: the FP16 node kernel with its fused row maxima) - the squared distances of a 200-node graph are where the
    per-row ranging of the edge activations has the most to do."""
    sd_np, sd, cfg, xh, nm, em = _oracle_case([200, 137], H, 2, seed=808, n_max=200)
    em = em.clone()
    em[1, :30, 100:137] = False                     # ligand fragments do not see the far half of this pocket
    em[1, 100:137, :30] = False
    t = torch.tensor([[0.2], [0.7]])
    with torch.no_grad():
        ref = orc.dynamics_forward(sd, cfg, t, xh, nm, em, None, 30, prefix="dynamics.egnn.")
    dyn = build_dynamics(sd_np, H, 2)
    dyn.precision = precision
    out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, 30).cpu()
    assert_parity(out.numpy(), ref.numpy(), f"pocket-sized graph {precision} H={H}")
    assert np.all(out.numpy()[~nm.numpy()[..., 0]] == 0.0)
    # nodes >= mol_shape are fixed: their velocity is that of a rigid translation only (the centre-of-gravity removal)
    v = out[0, 30:200, :3]
    assert float((v - v[0:1]).abs().max()) < 1e-6


@pytest.mark.parametrize("H", [32, 64, 128])
def test_fp16x3_node_path_by_width(H):
    """Which node kernels the fp16x3 mode runs, stated as behaviour: below width 128 the FP16 node kernel does not exist
    and the mode runs the exact-fp32 node kernels (k_node_f32 / k_gemm_r16; the edge kernels are FP16 at every width), from
    128 up it runs k_node<..., F16>.  With the second edge Linear, its bias and the coordinate head zeroed every message
    and every coordinate update is exactly 0 in every arithmetic (SiLU(0) = 0), so the output features are a function of
    the NODE path alone: bit-equal to the fp32 mode's at widths 32 and 64, fp32-accurate but not bit-equal at 128."""
    from hierdiff_amd.weights import synthetic_state_dict
    L = 2
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 717, 1.0)
    for k in list(sd_np):
        if ".edge_mlp.2." in k or ".coord_mlp.4." in k or ".coord_mlp.2." in k:
            sd_np[k] = np.zeros_like(sd_np[k])
    xh, nm, em = orc.random_inputs([30, 17, 1, 24, 9, 30], 8, 71, 30)
    t = torch.linspace(0.1, 0.9, 6).view(6, 1)
    out = {}
    for precision in ("fp32", "fp16x3"):
        dyn = build_dynamics(sd_np, H, L)
        dyn.precision = precision
        out[precision] = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu()
        assert torch.isfinite(out[precision]).all()
        assert float(out[precision][..., :3].abs().max()) == 0.0           # no coordinate update at all
    same = torch.equal(out["fp32"], out["fp16x3"])
    if H < 128:
        assert same, f"H={H}: fp16x3 is expected to run the fp32 node kernels (max diff {(out['fp32'] - out['fp16x3']).abs().max():.3e})"
    else:
        assert not same, "H=128: fp16x3 is expected to run the FP16 node kernel"
        assert rel_l2(out["fp16x3"].numpy(), out["fp32"].numpy()) < 2e-6


@pytest.mark.parametrize("H,L,B", [(64, 2, 210), (256, 1, 190), (32, 2, 190), (128, 2, 190)])
def test_fp32_node_paths_agree_bitwise(H, L, B):
    """fp32 mode runs the node side fused (k_node_f32: one launch per update, 32-row workgroups on 32 x 32 x 2 MFMAs) from
    5,400 active rows on and as three launches below: widths >= 128 k_node_split_f32 (32 x 32 output tiles, the four K quarters
    of a contraction on four wavefronts - k_node_f32 sums in the same quarters), narrower widths k_gemm_r16 (16-row workgroups on
    16 x 16 x 4 MFMAs, one chain like their fused kernel).  Small batches do not fill 256 CUs with a serial 50 us chain per row
    tile.  Each pair is bit-identical by construction (the same fmaf chains per output element, bias added after the
    contraction), so a molecule's bits do not depend on the size of the batch it is sampled in: B molecules (>= 5,700 rows,
    fused) against their first 8 alone (240 rows), narrow and production widths."""
    from hierdiff_amd.weights import synthetic_state_dict
    N = 30
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 515, 1.0)
    dyn = build_dynamics(sd_np, H, L)
    dyn.precision = "fp32"
    xh, nm, em = orc.random_inputs([N] * B, 8, 33)
    xh, nm = xh.to(DEV), nm.to(DEV)
    t = torch.linspace(0.05, 0.95, B, device=DEV).view(-1, 1)
    big = dyn._forward(t, xh, nm, None, None, None)
    small = dyn._forward(t[:8], xh[:8], nm[:8], None, None, None)
    assert torch.isfinite(big).all()
    assert torch.equal(big[:8], small)


@pytest.mark.parametrize("H,L,B", [(256, 2, 100), (128, 2, 100), (256, 1, 90)])
def test_fp16x3_node_paths_agree_bitwise(H, L, B):
    """fp16x3 runs the node update fused (k_node<..., F16>: one launch, one workgroup per 32 rows) from 2,048 active rows on and
    as the three launches of k_node_split below (32 x 32 output tiles over many workgroups, the four K quarters of a contraction
    on four wavefronts: a batch of 2 - 64 molecules does not fill 256 CUs with a serial 20 - 27 us chain per row tile).  Both sum
    every contraction as ((q0 + q1) + q2) + q3 over the same K quarters, range their FP16 operands by the same row bounds and
    leave the same row maxima for the edge kernels, so a molecule's bits do not depend on the size of the batch it is sampled
    in: B molecules (>= 2,600 rows, fused) against their first 30, 8 and 2 alone
    (k_node_split), ragged sizes."""
    from hierdiff_amd.weights import synthetic_state_dict
    N = 30
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 518, 1.0)
    dyn = build_dynamics(sd_np, H, L)
    dyn.precision = "fp16x3"
    n_list = [30, 17, 1, 24, 30, 9, 30, 28] + [30] * (B - 8)
    xh, nm, em = orc.random_inputs(n_list, 8, 34, N)
    xh, nm, em = xh.to(DEV), nm.to(DEV), em.to(DEV)
    t = torch.linspace(0.05, 0.95, B, device=DEV).view(-1, 1)
    big = dyn._forward(t, xh, nm, em, None, None)
    assert dyn.topology(nm, em, B, N).info()["nodes"] >= 2048
    assert torch.isfinite(big).all()
    for k in (30, 8, 2):
        small = dyn._forward(t[:k], xh[:k], nm[:k].contiguous(), em[:k].contiguous(), None, None)
        assert torch.equal(big[:k], small), f"k={k}: max diff {(big[:k] - small).abs().max().item():.3e}"


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("H,C_,general", [(256, 0, False), (128, 1, True)])
def test_small_batch_edge_kernel_is_bit_identical(H, C_, general, precision):
    """Topologies of at most 512 edge tiles (B <= 18 at N = 30: the reference's shipped job is batch_size 2) run through
    k_edge_split - one tile per workgroup, its columns over the four wavefronts - and larger ones through k_edge (one tile
    per wavefront), in every precision mode.  Same MFMA chain per output element, the row dot handed from wavefront to
    wavefront in column order, same gate / head / masked sums: the first molecules of a 40-molecule batch (k_edge) against
    the same molecules alone (k_edge_split), GCL and coordinate layers, ragged sizes, context, holes in the edge mask,
    torch.equal."""
    from hierdiff_amd.weights import synthetic_state_dict
    L, N = 2, 30
    sd_np = synthetic_state_dict(9, C_, H, L, 2, True, 615, 1.0)
    dyn = build_dynamics(sd_np, H, L, C_)
    dyn.precision = precision
    n_list = [30, 17, 1, 24, 30, 9] + [30] * 34
    xh, nm, em = orc.random_inputs(n_list, 8, 35, N)
    if general:
        em = em.clone().reshape(len(n_list), N, N)
        em[:, 3, 5] = False
        em[:, 7, 2] = False
    ctx = torch.randn(len(n_list), N, C_).to(DEV) if C_ else None
    xh, nm, em = xh.to(DEV), nm.to(DEV), em.to(DEV)
    t = torch.linspace(0.05, 0.95, len(n_list), device=DEV).view(-1, 1)
    big = dyn._forward(t, xh, nm, em, ctx, None)
    assert dyn.topology(nm, em, len(n_list), N).info()["tiles"] > 700
    for k in (1, 2, 6):
        nmk, emk = nm[:k].contiguous(), em[:k].contiguous()
        assert dyn.topology(nmk, emk, k, N).info()["tiles"] <= 512
        small = dyn._forward(t[:k], xh[:k], nmk, emk, None if ctx is None else ctx[:k], None)
        assert torch.isfinite(small).all()
        assert torch.equal(big[:k], small), f"k={k}: max diff {(big[:k] - small).abs().max().item():.3e}"


@pytest.mark.parametrize("precision", PRECISIONS)
def test_mixed_edge_launch_is_bit_identical(precision):
    """Above one whole-tile workgroup per CU (1,024 tiles), when few tiles are left over after the last full round, the edge
    layers run k_edge_mixed: a multiple of the CU count of whole-tile workgroups (k_edge's body) plus the left-over tiles as
    column-split workgroups (k_edge_split's body) that back-fill (BASELINE config 2's B = 64 in fp32).  The molecules of such
    batches against the same molecules at the head of a 150-molecule batch (3,928 tiles: plain k_edge) and the first six alone
    (k_edge_split): torch.equal, ragged sizes, every precision mode."""
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, N = 256, 2, 30
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 616, 1.0)
    dyn = build_dynamics(sd_np, H, L, 0)
    dyn.precision = precision
    n_list = ([30, 17, 1, 24, 30, 9] + [30] * 26) * 2 + [30] * 86
    xh, nm, em = orc.random_inputs(n_list, 8, 36, N)
    xh, nm, em = xh.to(DEV), nm.to(DEV), em.to(DEV)
    t = torch.linspace(0.05, 0.95, len(n_list), device=DEV).view(-1, 1)
    big = dyn._forward(t, xh, nm, em, None, None)
    assert dyn.topology(nm, em, len(n_list), N).info()["tiles"] > 3072
    # 46 molecules: 1,092 tiles = 256 whole-tile workgroups + 68 column-split tiles (mixed in every mode); 64 molecules:
    # 1,584 tiles, 560 left over (mixed in fp32 only: the rule of launch_edge_h)
    for k in (46, 64):
        nmk, emk = nm[:k].contiguous(), em[:k].contiguous()
        tiles = dyn.topology(nmk, emk, k, N).info()["tiles"]
        assert 1024 < tiles < 3072, tiles
        mid = dyn._forward(t[:k], xh[:k], nmk, emk, None, None)
        assert torch.isfinite(mid).all()
        assert torch.equal(big[:k], mid), f"k={k}: max diff {(big[:k] - mid).abs().max().item():.3e}"
    nm6, em6 = nm[:6].contiguous(), em[:6].contiguous()
    small = dyn._forward(t[:6], xh[:6], nm6, em6, None, None)
    assert torch.equal(mid[:6], small)


def test_equivariance_permutation_padding_full_size():
    """Size-independent properties at the headline shape B=256, N=30, H=256, L=6."""
    from hierdiff_amd.weights import synthetic_state_dict
    H, L, B, N = 256, 6, 256, 30
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 123, 1.0)
    dyn = build_dynamics(sd_np, H, L)
    xh, nm, em = orc.random_inputs([N] * B, 8, 9)
    xh, nm = xh.to(DEV), nm.to(DEV)
    t = torch.full((B, 1), 0.4, device=DEV)
    out = dyn._forward(t, xh, nm, em.to(DEV), None, None)
    assert torch.isfinite(out).all()
    # O(3): rotate inputs -> velocity rotates, features invariant
    g = torch.Generator().manual_seed(3)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    q = q.to(DEV)
    xr = torch.cat([xh[..., :3] @ q, xh[..., 3:]], dim=-1)
    outr = dyn._forward(t, xr, nm, em.to(DEV), None, None)
    assert rel_l2((out[..., :3] @ q).cpu().numpy(), outr[..., :3].cpu().numpy()) < 1e-4
    assert rel_l2(out[..., 3:].cpu().numpy(), outr[..., 3:].cpu().numpy()) < 1e-5
    # permutation of nodes inside each molecule
    perm = torch.randperm(N, generator=g).to(DEV)
    outp = dyn._forward(t, xh[:, perm], nm, em.to(DEV), None, None)
    assert rel_l2(out[:, perm].cpu().numpy(), outp.cpu().numpy()) < 1e-5
    # padding invariance: same molecules padded to N=40
    pad = torch.zeros(B, 10, xh.shape[2], device=DEV)
    xh40 = torch.cat([xh, pad], dim=1)
    nm40 = torch.cat([nm, torch.zeros(B, 10, 1, dtype=torch.bool, device=DEV)], dim=1)
    out40 = dyn.forward_with_topology(dyn.topology(nm40, None, B, 40), t, xh40, None, None)
    assert torch.equal(out40[:, 30:], torch.zeros_like(out40[:, 30:]))
    assert rel_l2(out40[:, :30].cpu().numpy(), out.cpu().numpy()) < 1e-6
    # centre of gravity of the velocity
    assert out[..., :3].sum(1).abs().max().item() < 1e-4
    # run-to-run bit reproducibility
    assert torch.equal(out, dyn._forward(t, xh, nm, em.to(DEV), None, None))


def test_philox_device_matches_host_twin_and_is_shard_independent():
    from hierdiff_amd import _lib
    import ctypes as C
    from hierdiff_amd.weights import synthetic_state_dict
    lib = _lib.load()
    sd_np = synthetic_state_dict(9, 0, 32, 1)
    model = build_diffusion(sd_np, 32, 1, T=4)
    B, N = 6, 5
    nm = torch.ones(B, N, 1, dtype=torch.bool, device=DEV)
    h = model._lib_handle()
    topo = model.dynamics.topology(nm, None, B, N)
    z = torch.empty(B, N, 11, device=DEV)
    _lib.check(lib.hd_noise(h, topo.ptr, None, None, B, 99, 10, 2, 0, z.data_ptr(), 0))
    torch.cuda.synchronize()
    raw = np.array([[lib.hd_philox_normal_host(99, 10 + b, 2, i) for i in range(N * 11)] for b in range(B)],
                   dtype=np.float32).reshape(B, N, 11)
    exp = raw.copy()
    exp[:, :, :3] -= raw[:, :, :3].mean(1, keepdims=True)
    np.testing.assert_allclose(z.cpu().numpy(), exp, rtol=0, atol=3e-6)
    # rows depend on the global sample id only: shard [3:6) of the batch drawn alone is identical
    nm2 = torch.ones(3, N, 1, dtype=torch.bool, device=DEV)
    topo2 = model.dynamics.topology(nm2, None, 3, N)
    z2 = torch.empty(3, N, 11, device=DEV)
    _lib.check(lib.hd_noise(h, topo2.ptr, None, None, 3, 99, 13, 2, 0, z2.data_ptr(), 0))
    torch.cuda.synchronize()
    assert torch.equal(z2, z[3:])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_sample_loop_graph_equals_plain_and_shards_reproduce(precision):
    """Full-length (T = 1000) sampling: hipGraph replay == plain launches, repeated calls reuse the cached graph, and
    every shard of the batch - the samples a rank would own under hierdiff_amd.sharding - reproduces its rows of the
    full batch BIT FOR BIT: edge tiles are cut per molecule (hd_topology_create), the noise is
    keyed by the global sample id."""
    from hierdiff_amd.weights import synthetic_state_dict
    sd_np = synthetic_state_dict(9, 0, 64, 2, 2, True, 5, 1.0)
    T = 1000
    model = build_diffusion(sd_np, 64, 2, T=T, precision=precision)
    n_list = [7, 3, 9, 5, 1, 8, 12, 2, 6, 9, 4]
    nm, _ = orc.canonical_masks(n_list)
    nm = nm.to(DEV)
    model.use_graph = False
    x0, h0 = model.sample_from_masks(nm, None, None, sample_id_base=100)
    model.use_graph = True
    x1, h1 = model.sample_from_masks(nm, None, None, sample_id_base=100)
    assert torch.equal(x0, x1) and torch.equal(h0, h1)
    x2, h2 = model.sample_from_masks(nm, None, None, sample_id_base=100)        # cached graph, fresh z tensors
    assert torch.equal(x0, x2) and torch.equal(h0, h2)
    assert torch.isfinite(x0).all() and torch.isfinite(h0).all()
    x3, _ = model.sample_from_masks(nm, None, None, sample_id_base=500)         # same graph, other global ids
    assert not torch.equal(x3, x0)
    from hierdiff_amd.sharding import shard_sample_ids
    for world in (2, 3, 8):
        for rank in range(world):
            lo, cnt = shard_sample_ids(0, len(n_list), rank, world)
            if cnt == 0:
                continue
            sub = n_list[lo:lo + cnt]
            nm_s, _ = orc.canonical_masks(sub)          # the shard pads to ITS largest molecule
            xs, hs = model.sample_from_masks(nm_s.to(DEV), None, None, sample_id_base=100 + lo)
            n2 = nm_s.shape[1]
            assert torch.equal(xs, x0[lo:lo + cnt, :n2]) and torch.equal(hs, h0[lo:lo + cnt, :n2]), (world, rank)


def test_public_sample_api_result_format():
    from hierdiff_amd.weights import synthetic_state_dict
    sd_np = synthetic_state_dict(9, 0, 32, 1, 2, True, 6, 1.0)
    model = build_diffusion(sd_np, 32, 1, T=5)
    torch.manual_seed(0)
    res, names = model.sample_batches(4, 2, DEV)
    assert len(res) == 8 and names == []
    for r in res:
        n = r["x"].shape[0]
        assert r["x"].shape == (n, 3) and r["h"].shape == (n, 8)
        assert r["x"].device.type == "cpu" and r["x"].dtype == torch.float32
        assert torch.isfinite(r["x"]).all() and torch.isfinite(r["h"]).all()
    # torch-RNG mode walks the reference's per-step API and must agree with the fused loop's maths:
    model.noise_mode = "torch"
    torch.manual_seed(1)
    res2 = model.sample(3, DEV)
    assert len(res2) == 3 and all(torch.isfinite(r["x"]).all() for r in res2)


def test_edm_signature_fix_noise():
    from hierdiff_amd import EnVariationalDiffusion, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    sd_np = synthetic_state_dict(9, 1, 32, 1, 2, True, 8, 1.0)
    m = EnVariationalDiffusion(default_config(hidden_nf=32, n_layers=1, context_node_nf=1, timesteps=6))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.to(DEV)
    nm, em = orc.canonical_masks([5, 5, 5])
    ctx = torch.full((3, 5, 1), 1.7)
    x, h = m.sample(3, 5, nm.to(DEV), em.to(DEV), ctx.to(DEV), fix_noise=True)
    # identical masks + shared noise + same context => identical samples across the batch
    assert torch.allclose(x[0], x[1], atol=1e-5) and torch.allclose(h[0], h[2], atol=1e-5)
    assert x.shape == (3, 5, 3) and h.shape == (3, 5, 8)


def test_cli_sampler_writes_reference_wire_format(tmp_path):
    """hierdiff_amd.sampler (replacement of endiffusion/sampler.py): checkpoint in, sample_results.pkl out."""
    import pickle
    from hierdiff_amd import sampler
    from hierdiff_amd.weights import synthetic_state_dict
    syn = synthetic_state_dict(9, 0, 32, 1, 2, True, 12, 1.0)
    ck = tmp_path / "diffusion.ckpt"
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v.copy()) for k, v in syn.items()}}, ck)
    out = tmp_path / "sample_results.pkl"
    rc = sampler.main(["--checkpoint", str(ck), "--batch-size", "3", "--num-batches", "2", "--out", str(out),
                       "--hidden-nf", "32", "--n-layers", "1", "--timesteps", "4"])
    assert rc == 0
    with open(out, "rb") as f:
        results, names = pickle.load(f)
    assert len(results) == 6 and names == []
    for r in results:
        n = r["x"].shape[0]
        assert 1 <= n <= 83 and r["x"].shape == (n, 3) and r["h"].shape == (n, 8)
        assert torch.isfinite(r["x"]).all() and torch.isfinite(r["h"]).all()
        assert r["x"].sum(0).abs().max() < 1e-3 * max(1.0, r["x"].abs().max().item()) * n   # centre of gravity ~ 0
    # same seed -> same molecules (counter-based noise, seeded node counts)
    out2 = tmp_path / "again.pkl"
    sampler.main(["--checkpoint", str(ck), "--batch-size", "3", "--num-batches", "2", "--out", str(out2),
                  "--hidden-nf", "32", "--n-layers", "1", "--timesteps", "4"])
    with open(out2, "rb") as f:
        again = pickle.load(f)[0]
    assert all(torch.equal(a["x"], b["x"]) and torch.equal(a["h"], b["h"]) for a, b in zip(results, again))


def test_cli_sampler_from_reference_yaml_configs(tmp_path):
    """The CLI pointed at a config tree laid out like the reference's `endiffusion/conf` gives the molecules of the keyword route."""
    import pickle
    from hierdiff_amd import sampler
    from hierdiff_amd.weights import synthetic_state_dict
    from tests.test_cabi_cpu import write_reference_style_conf
    conf = write_reference_style_conf(tmp_path, H=32, L=1, T=4, batch_size=3, num_batches=2, with_analyze=False)
    syn = synthetic_state_dict(9, 0, 32, 1, 2, True, 12, 1.0)
    ck = tmp_path / "diffusion.ckpt"
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v.copy()) for k, v in syn.items()}}, ck)
    a, b = tmp_path / "a.pkl", tmp_path / "b.pkl"
    assert sampler.main(["--checkpoint", str(ck), "--out", str(a), "--model-config", str(conf / "model" / "ddpmgblur.yaml"),
                         "--sample-config", str(conf / "sample" / "default.yaml")]) == 0
    assert sampler.main(["--checkpoint", str(ck), "--out", str(b), "--batch-size", "3", "--num-batches", "2",
                         "--hidden-nf", "32", "--n-layers", "1", "--timesteps", "4"]) == 0
    ra, rb = pickle.load(open(a, "rb"))[0], pickle.load(open(b, "rb"))[0]
    assert len(ra) == len(rb) == 6
    assert all(torch.equal(p["x"], q["x"]) and torch.equal(p["h"], q["h"]) for p, q in zip(ra, rb))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_pocket_conditioned_chain_golden(precision):
    """F8 (diffusion_qm9.py:362-382): residue nodes ride along as fixed rows behind the molecule."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    fx = load("f8_pocket_h64_l2")
    H, L, T = int(fx["hidden_nf"]), int(fx["n_layers"]), int(fx["T"])
    cfg = default_config(hidden_nf=H, n_layers=L, timesteps=T)
    cfg["pocket"] = True
    model = DiffusionQM9(cfg)
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, int(fx["weight_seed"]), float(fx["coord_gain"]), pocket=True)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    model = model.to(DEV)
    model.dynamics.precision = precision
    model.schedule_gammas = fx["gamma_grid"]
    n_list = [int(v) for v in fx["n_list"]]
    nm, em = orc.canonical_masks(n_list)
    raws = [(torch.from_numpy(fx["raw_x"][i]), torch.from_numpy(fx["raw_h"][i])) for i in range(T + 2)]
    feat = model.pocket_embed(torch.from_numpy(fx["pocket_feat"]).to(DEV).long())
    pocket = (torch.from_numpy(fx["pocket_pos"]).to(DEV), feat, torch.from_numpy(fx["pocket_node_mask"]).to(DEV),
              torch.from_numpy(fx["pocket_edge_mask"]).to(DEV))
    with torch.no_grad():
        x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws, pocket=pocket)
    assert_parity(x.cpu().numpy() * nm.float().numpy(), fx["x"], "pocket x")
    assert_parity(h.cpu().numpy(), fx["h"], "pocket h")


def test_pocket_public_api():
    """sample(pocket_cond=...) and sample_batches(protein_data_all=...) as the reference exposes them."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    cfg = default_config(hidden_nf=32, n_layers=1, timesteps=4)
    cfg["pocket"] = True
    model = DiffusionQM9(cfg)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in
                           synthetic_state_dict(9, 0, 32, 1, 2, True, 3, 1.0, pocket=True).items()})
    model = model.to(DEV)
    rng = np.random.default_rng(0)
    prot = [{"residue_type": ["ALA", "GLY", "TRP", "SER", "VAL"][: 3 + (k % 3)], "coord": rng.normal(size=(3 + (k % 3), 3)).tolist(),
             "pocket_name": f"p{k}", "ligand_name": f"l{k}"} for k in range(4)]
    torch.manual_seed(0)
    res, names = model.sample_batches(2, 2, DEV, protein_data_all=prot)
    assert len(res) == 4 and isinstance(names, list)
    for r in res:
        assert r["x"].shape[1] == 3 and r["h"].shape[1] == 8 and torch.isfinite(r["x"]).all()
    # a model without pocket support refuses pocket_cond
    plain = build_diffusion(synthetic_state_dict(9, 0, 32, 1), 32, 1, T=3)
    from hierdiff_amd.diffusion import pocket_tensors
    with pytest.raises(ValueError):
        plain.sample(2, DEV, pocket_cond=[t[:2] for t in pocket_tensors(prot)])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_full_length_chain_vs_oracle(precision):
    """The workload's real length: T = 1000 posterior steps + decode with injected normals, HIP path vs the CPU
    oracle (H=32, L=2 so the oracle's 1001 forwards take well under a minute).  Trajectory-level bar: rel-L2 < 1e-3
    on the final x and h (the per-forward bar stays 1e-4; measured here 2e-5 / 5e-6 in fp32)."""
    import copy
    from hierdiff_amd.weights import synthetic_state_dict
    from hierdiff_amd.noise_model import evaluate_gamma
    H, L, T = 32, 2, 1000
    n_list = [8, 5, 7, 3]
    sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 5, 1.0)
    cfg = orc.DynCfg(in_node_nf=9, context_node_nf=0, hidden_nf=H, n_layers=L, normalization_factor=10.0)
    nm, em = orc.canonical_masks(n_list)
    B, N = nm.shape[:2]
    g = torch.Generator().manual_seed(11)
    raws = [(torch.randn(B, N, 3, generator=g), torch.randn(B, N, 8, generator=g)) for _ in range(T + 2)]
    model = build_diffusion(sd_np, H, L, T=T, precision=precision)
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws)
    key = ("full_chain_oracle", H, L, T)
    if key not in _CACHE:       # the oracle replays the gamma table the product evaluates (fp64 on the host, rounded once)
        gg = evaluate_gamma(copy.deepcopy(model.gamma).cpu(), (torch.arange(T + 1, dtype=torch.float64) / T).view(-1, 1)).view(-1)
        _CACHE[key] = orc.sample_chain(orc.as_torch_sd(sd_np), cfg, T, nm, em, None, raws, gamma_grid=gg)
    xo, ho = _CACHE[key]
    nmf = nm.float().numpy()
    rx = rel_l2(x.cpu().numpy() * nmf, xo.numpy() * nmf)
    rh = rel_l2(h.cpu().numpy(), ho.numpy())
    print(f"T=1000 chain [{precision}]: x {rx:.2e} h {rh:.2e}")
    assert rx < 1e-3 and rh < 1e-3
    assert torch.isfinite(x).all() and torch.isfinite(h).all()


_CACHE = {}


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["f9_nll_eval_h64_l2", "f9_nll_train_h64_l2"])
def test_nll_forward_golden(name, precision):
    """DiffusionQM9.compute_loss / nll / forward(batch) value (validation NLL: two network calls with per-row t;
    training-mode value: one call incl. the t == 0 branch) against the reference, with its draws and schedule values
    replayed.  Tolerance: 1e-4 relative per molecule (+1e-3 absolute), the per-forward bar carried through the loss."""
    fx = load(name)
    sd_np, _, _ = fixture_model(fx)
    model = build_diffusion(sd_np, int(fx["hidden_nf"]), int(fx["n_layers"]), T=int(fx["T"]), precision=precision)
    training = bool(int(fx["training"]))
    model.train(training)
    nm, em = orc.canonical_masks([int(v) for v in fx["n_list"]])
    x, h = torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["h"]).to(DEV)
    gam = {k: fx[k] for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
    replay = dict(t_int=fx["t_int"], eps=fx["eps"], gammas=gam)
    if not training:
        replay["eps0"] = fx["eps0"]
    loss, info = model.compute_loss(x, h, nm.to(DEV), em.to(DEV), None, t0_always=not training, **replay)
    np.testing.assert_allclose(loss.cpu().numpy(), fx["loss"], rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(info["error"].cpu().numpy(), fx["error"], rtol=1e-4, atol=1e-4)
    # the batch-level entry point of the reference's training / validation step
    B, N = x.shape[:2]
    batch = {"positions": x, "atom_mask": nm.to(DEV), "edge_mask": em.to(DEV).view(B, N, N), "node_feature": h}
    out = model(batch, **replay)
    assert abs(out["loss"].item() - float(np.mean(fx["loss"]))) <= 1e-4 * abs(float(np.mean(fx["loss"]))) + 1e-3
    # without replay the draws come from torch's generator: finite and reproducible under a seed
    torch.manual_seed(5); a = model.nll(x, h, nm.to(DEV), em.to(DEV))
    torch.manual_seed(5); b = model.nll(x, h, nm.to(DEV), em.to(DEV))
    assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_norm_values_chain_and_nll_golden(precision):
    """F20: non-unit norm_values / norm_biases - `unnormalize` behind the library's decode kernel (sampling chain with the reference's
    noise and schedule values replayed) and `nll` on raw data in evaluation and training mode."""
    from hierdiff_amd import DiffusionQM9, default_config
    fx = load("f20_norm_h64_l2")
    sd_np, _, _ = fixture_model(fx)
    nv = [float(v) for v in fx["norm_values"]]
    nb = [None] + [float(v) for v in fx["norm_biases"][1:]]
    n_list = [int(v) for v in fx["n_list"]]
    nm, em = orc.canonical_masks(n_list)

    def build(T):
        cfg = default_config(hidden_nf=int(fx["hidden_nf"]), n_layers=int(fx["n_layers"]), timesteps=T)
        cfg.norm_values, cfg.norm_biases = nv, nb
        m = DiffusionQM9(cfg)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()})
        m.dynamics.precision = precision
        return m.to(DEV)
    T = int(fx["T_chain"])
    model = build(T)
    model.schedule_gammas = fx["gamma_grid"]
    raws = [(torch.from_numpy(fx["raw_x"][i]), torch.from_numpy(fx["raw_h"][i])) for i in range(T + 2)]
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws)
    nmf = nm.float().numpy()
    tol = 1e-4
    assert_parity(x.cpu().numpy() * nmf, fx["chain_x"], "F20 chain x", tol, 10 * tol)
    assert_parity(h.cpu().numpy(), fx["chain_h"], "F20 chain h", tol, 10 * tol)
    model = build(int(fx["T"]))
    xr, hr = torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["h"]).to(DEV)
    for tag, training in (("eval", False), ("train", True)):
        model.train(training)
        gam = {k: fx[f"{tag}_{k}"] for k in ("gamma_s", "gamma_t", "gamma_0", "gamma_T")}
        replay = dict(t_int=fx[f"{tag}_t_int"], eps=fx[f"{tag}_eps"], gammas=gam)
        if not training:
            replay["eps0"] = fx["eval_eps0"]
        loss = model.nll(xr, hr, nm.to(DEV), em.to(DEV), None, **replay)
        np.testing.assert_allclose(loss.cpu().numpy(), fx[f"{tag}_nll"], rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["f21_gnn_h64_l3", "f21_gnn_h256_l2_mean"])
def test_gnn_dynamics_golden(name, precision):
    """F21: mode 'gnn_dynamics' on the library's kernels (an internal egnn-mode engine: coordinates fed as features, all-pairs edge
    mask so that padded nodes and self pairs send messages as in the reference, which passes no edge mask there)."""
    from hierdiff_amd import EGNN_dynamics_QM9
    from tests.test_oracle_golden import _gnn_case
    fx = load(name)
    sd_np, cfg = _gnn_case(fx)
    dyn = EGNN_dynamics_QM9(9, 0, 3, hidden_nf=cfg.hidden_nf, n_layers=cfg.n_layers, attention=cfg.attention, mode="gnn_dynamics",
                            normalization_factor=cfg.normalization_factor, aggregation_method=cfg.aggregation_method)
    assert list(dyn.state_dict().keys()) == list(sd_np.keys())
    dyn.load_numpy_state_dict(sd_np)
    dyn = dyn.to(DEV).eval()
    dyn.precision = precision
    xh, nm = torch.from_numpy(fx["xh"]).to(DEV), torch.from_numpy(fx["node_mask"]).to(DEV)
    B, N = xh.shape[:2]
    em = torch.zeros(B, N * N, dtype=torch.bool, device=DEV)          # whatever the caller passes: the reference ignores it in this mode
    out = dyn._forward(torch.from_numpy(fx["t_rows"]).to(DEV), xh, nm, em, None)
    assert_parity(out.cpu().numpy(), fx["out_row_t"], name + " row t")
    out = dyn._forward(torch.from_numpy(fx["t_scalar"]).to(DEV), xh, nm, em, None)
    assert_parity(out.cpu().numpy(), fx["out_scalar_t"], name + " scalar t")
    assert np.all(out.cpu().numpy()[~fx["node_mask"][..., 0]] == 0.0)
    # new weights are picked up (the engine's copy is keyed on the parameters' versions)
    with torch.no_grad():
        dyn.gnn.embedding_out.bias.add_(1.0)
    out2 = dyn._forward(torch.from_numpy(fx["t_scalar"]).to(DEV), xh, nm, em, None)
    assert not torch.equal(out, out2)
    # inference only; the sampler loop / training path stay egnn_dynamics'
    dyn.train()
    with torch.enable_grad(), pytest.raises(NotImplementedError):
        dyn._forward(torch.from_numpy(fx["t_scalar"]).to(DEV), xh, nm, em, None)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_gnn_dynamics_sampling_chain_golden(precision):
    """F21c: DiffusionQM9 built with dynamics.mode = 'gnn_dynamics' samples through its step-by-step loop (network through the gnn
    `_forward`, posterior step / decode kernels of the library): the reference's own chain with its noise and schedule values."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.weights import synthetic_gamma_state_dict, synthetic_gnn_state_dict
    fx = load("f21c_gnn_chain_h64_l2")
    H, L, seed, T = int(fx["hidden_nf"]), int(fx["n_layers"]), int(fx["weight_seed"]), int(fx["T"])
    cfg = default_config(hidden_nf=H, n_layers=L, timesteps=T)
    cfg.dynamics.mode = "gnn_dynamics"
    model = DiffusionQM9(cfg)
    sd_np = {"gamma." + k: v for k, v in synthetic_gamma_state_dict(seed).items()}
    sd_np.update({"dynamics." + k: v for k, v in synthetic_gnn_state_dict(9, 0, H, L, True, seed).items()})
    sd_np["buffer"] = np.zeros(1, np.float32)
    assert sorted(model.state_dict().keys()) == sorted(sd_np.keys())
    model.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd_np.items()})
    model = model.to(DEV).eval()
    model.dynamics.precision = precision
    model.schedule_gammas = fx["gamma_grid"]
    nm, em = orc.canonical_masks([int(v) for v in fx["n_list"]])
    raws = [(torch.from_numpy(fx["raw_x"][i]), torch.from_numpy(fx["raw_h"][i])) for i in range(T + 2)]
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws)
    tol = 1e-4
    assert_parity(x.cpu().numpy() * nm.float().numpy(), fx["x"], "F21c x", tol, 10 * tol)
    assert_parity(h.cpu().numpy(), fx["h"], "F21c h", tol, 10 * tol)
    # the public entry point (torch-generator noise in this mode): finite molecules of the drawn sizes
    torch.manual_seed(3)
    res = model.sample(4, DEV)
    assert len(res) == 4 and all(torch.isfinite(r["x"]).all() and r["x"].shape[1] == 3 and r["h"].shape[1] == 8 for r in res)


def test_c_abi_error_codes_and_messages():
    """Error behaviour of the boundary (include/hierdiff_hip.h): negative codes + hd_last_error(), no aborts;
    the Python mirror turns them into HierDiffHipError / ValueError."""
    import ctypes as C
    from hierdiff_amd import _lib
    lib = _lib.load()
    last = lambda: lib.hd_last_error().decode()

    def cfg(**kw):
        base = dict(in_node_nf=9, context_node_nf=0, n_dims=3, hidden_nf=32, n_layers=1, inv_sublayers=1, attention=1,
                    tanh=1, condition_time=1, norm_constant=0.0, normalization_factor=10.0, coords_range=30.0, precision=3)
        base.update(kw)
        return _lib.HdConfig(**base)
    h = C.c_void_p()
    assert lib.hd_create(C.byref(cfg(hidden_nf=48)), 0, C.byref(h)) == -1 and "hidden_nf" in last()
    assert lib.hd_create(C.byref(cfg(precision=7)), 0, C.byref(h)) == -1 and "precision" in last()
    assert lib.hd_create(C.byref(cfg(normalization_factor=0.0)), 0, C.byref(h)) == -1
    assert lib.hd_create(C.byref(cfg()), 99, C.byref(h)) == -2 and "device" in last()
    assert lib.hd_create(C.byref(cfg()), 0, C.byref(h)) == 0
    n = lib.hd_weight_count(h)
    w = torch.zeros(n, device=DEV)
    assert lib.hd_set_weights(h, C.c_void_p(w.data_ptr()), n - 1, 1, None) == -1 and "expected" in last()
    nm = np.ones((2, 4), dtype=np.uint8)
    topo = C.c_void_p()
    assert lib.hd_topology_create(h, nm.ctypes.data_as(C.c_void_p), None, 0, 4, C.byref(topo)) == -1
    assert lib.hd_topology_create(h, nm.ctypes.data_as(C.c_void_p), None, 2, 4, C.byref(topo)) == 0
    xh = torch.zeros(2, 4, 11, device=DEV); out = torch.zeros_like(xh); t = torch.zeros(2, device=DEV)
    args = lambda hh, tp, tn: (hh, tp, C.c_void_p(xh.data_ptr()), C.c_void_p(t.data_ptr()), tn, None, -1,
                               C.c_void_p(out.data_ptr()), None)
    assert lib.hd_egnn_forward(*args(h, topo, 2)) == -4 and "weights" in last()          # state error: no weights yet
    assert lib.hd_set_weights(h, C.c_void_p(w.data_ptr()), n, 1, None) == 0
    assert lib.hd_egnn_forward(*args(h, topo, 3)) == -1                                   # t must have 1 or B elements
    assert lib.hd_egnn_forward(h, topo, None, C.c_void_p(t.data_ptr()), 2, None, -1, C.c_void_p(out.data_ptr()), None) == -1
    h2 = C.c_void_p()
    assert lib.hd_create(C.byref(cfg()), 0, C.byref(h2)) == 0
    assert lib.hd_set_weights(h2, C.c_void_p(w.data_ptr()), n, 1, None) == 0
    assert lib.hd_egnn_forward(*args(h2, topo, 2)) == -1 and "another handle" in last()
    assert lib.hd_egnn_forward(*args(h, topo, 2)) == 0                                    # and the valid call still works
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert lib.hd_sample_loop(h, topo, C.c_void_p(xh.data_ptr()), None, 4, 3, 0, None, None, 0, 1, 0, 0, None) < 0     # no schedule set
    lib.hd_topology_destroy(topo); lib.hd_destroy(h); lib.hd_destroy(h2)
    # Python mirror
    from hierdiff_amd import EGNN_dynamics_QM9
    dyn = EGNN_dynamics_QM9(9, 0, 3, hidden_nf=32, n_layers=1).to(DEV)
    with pytest.raises((ValueError, _lib.HierDiffHipError)):
        dyn._forward(torch.zeros(3, 1, device=DEV), xh, torch.ones(2, 4, 1, dtype=torch.bool, device=DEV),
                     torch.ones(2, 16, dtype=torch.bool, device=DEV), None, None)
