"""Stage-2 layer E_GCL (/root/reference/models/egnn/gcl.py): oracle vs reference-generated golden vectors (CPU tier) and
the HIP path (`hd_egcl_forward` through hierdiff_amd.stage2.E_GCL) vs the same vectors and vs the oracle (GPU tier)."""
import numpy as np
import pytest
import torch

from oracle import egnn_oracle as orc
from tests.helpers import assert_parity, load

FIXTURES = ["f15_egcl_full_h64", "f15_egcl_full_h256", "f15_egcl_focal_h64", "f15_egcl_edge_h64", "f15_egcl_ctx_h64",
            "f15_egcl_geo_h64"]


def _case(fx):
    from hierdiff_amd.stage2 import synthetic_egcl_state_dict
    H, De, ctx = int(fx["hidden_nf"]), int(fx["edges_in_d"]), int(fx["context_nf"])
    att, eu = bool(int(fx["attention"])), bool(int(fx["edge_update"]))
    sd_np = synthetic_egcl_state_dict(H, De, ctx, att, eu, int(fx["weight_seed"]), coord_gain=0.3)
    cfg = orc.EGCLCfg(hidden_nf=H, edges_in_d=De, context_nf=ctx, attention=att, edge_update=eu, geo=bool(int(fx.get("geo", 0))))
    nm = torch.from_numpy(fx["node_mask"]) if int(fx["masked"]) else None
    em = torch.from_numpy(fx["edge_mask"]) if int(fx["has_edge_mask"]) else None
    return sd_np, cfg, nm, em


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_matches_reference(name):
    fx = load(name)
    sd_np, cfg, nm, em = _case(fx)
    with torch.no_grad():
        h, x, ea = orc.e_gcl_forward(orc.as_torch_sd(sd_np), cfg, fx["h"], torch.from_numpy(fx["row"]), torch.from_numpy(fx["col"]),
                                     fx["x"], fx["edge_attr"], nm, em)
    assert_parity(h.numpy(), fx["h_out"], name + " h", 2e-6, 2e-5)
    assert_parity(x.numpy(), fx["x_out"], name + " x", 2e-6, 2e-5)
    if cfg.edge_update:
        assert_parity(ea.numpy(), fx["edge_attr_out"], name + " edge_attr", 2e-6, 2e-5)
    else:
        assert ea is None


def test_module_mirrors_reference_layout():
    from hierdiff_amd.stage2 import E_GCL, egcl_param_shapes, synthetic_egcl_state_dict
    m = E_GCL(64, 64, 64, context_nf=0, edges_in_d=64, attention=True, tanh=True, coords_range=30, edge_update=True)
    shapes = egcl_param_shapes(64, 64, 0, True, True)
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_egcl_state_dict(64, 64, 0, True, True, 1).items()})
    m2 = E_GCL(64, 64, 64, edges_in_d=1, attention=False, tanh=True, coords_range=30, edge_update=False)
    assert "edge_mlp.0.weight" not in m2.state_dict() and m2.state_dict()["mes_mlp.0.weight"].shape == (64, 130)
    assert E_GCL(64, 64, 64, edges_in_d=64, geo=True)._cfg.geo == 1          # round 3: supported (fixture f15_egcl_geo_h64)
    for kw in (dict(agg="mean"), dict(angle_net=True), dict(edges_in_d=40)):
        with pytest.raises(NotImplementedError):
            E_GCL(64, 64, 64, **kw)
    with pytest.raises(NotImplementedError):
        E_GCL(64, 32, 64)
    from hierdiff_amd import _lib
    if _lib.load().hd_device_count() == 0:
        with pytest.raises(_lib.HierDiffHipError):
            m(torch.zeros(3, 64), [torch.tensor([0, 1]), torch.tensor([1, 2])], torch.zeros(3, 3), torch.zeros(2, 64))


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_hip_layer_golden(name):
    from hierdiff_amd.stage2 import E_GCL
    fx = load(name)
    sd_np, cfg, nm, em = _case(fx)
    dev = "cuda:0"
    m = E_GCL(cfg.hidden_nf, cfg.hidden_nf, cfg.hidden_nf, context_nf=cfg.context_nf, edges_in_d=cfg.edges_in_d,
              attention=cfg.attention, tanh=True, coords_range=30, edge_update=cfg.edge_update, geo=cfg.geo)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    m = m.to(dev)
    row, col = torch.from_numpy(fx["row"]).long().to(dev), torch.from_numpy(fx["col"]).long().to(dev)
    out = m(torch.from_numpy(fx["h"]).to(dev), [row, col], torch.from_numpy(fx["x"]).to(dev),
            edge_attr=torch.from_numpy(fx["edge_attr"]).to(dev), node_mask=None if nm is None else nm.to(dev),
            edge_mask=None if em is None else em.to(dev))
    assert len(out) == (3 if cfg.edge_update else 2)
    assert_parity(out[0].cpu().numpy(), fx["h_out"], name + " h")
    assert_parity(out[1].cpu().numpy(), fx["x_out"], name + " x")
    if cfg.edge_update:
        assert_parity(out[2].cpu().numpy(), fx["edge_attr_out"], name + " edge_attr")
    again = m(torch.from_numpy(fx["h"]).to(dev), [row, col], torch.from_numpy(fx["x"]).to(dev),
              edge_attr=torch.from_numpy(fx["edge_attr"]).to(dev), node_mask=None if nm is None else nm.to(dev),
              edge_mask=None if em is None else em.to(dev))
    assert all(torch.equal(a, b) for a, b in zip(out, again)), "deterministic (CSR sums, no atomics)"


@pytest.mark.gpu
def test_hip_stack_vs_oracle_beam_sized():
    """Three gcl_full layers chained as Edge_denoise does (edge_denoise.py:107-108) on a beam-sized dense batch
    (bs=24 graphs of 12 nodes, H=256, E = 3,456 edges incl. masked ones) against the oracle."""
    from hierdiff_amd.stage2 import E_GCL, synthetic_egcl_state_dict
    H, bs, n, dev = 256, 24, 12, "cuda:0"
    rng = np.random.Generator(np.random.PCG64(5))
    sizes = rng.integers(3, n + 1, size=bs)
    nm = torch.from_numpy((np.arange(n)[None, :] < sizes[:, None]).astype(np.float32)).reshape(bs * n, 1)
    ar = torch.arange(n)
    row = (ar.repeat_interleave(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n))
    col = (ar.repeat(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n))
    em = nm[row] * nm[col] * (row != col).float().unsqueeze(1)
    h = torch.from_numpy(rng.standard_normal((bs * n, H)).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((bs * n, 3)).astype(np.float32)) * nm
    ea = torch.from_numpy(rng.standard_normal((row.numel(), H)).astype(np.float32))
    cfg = orc.EGCLCfg(hidden_nf=H, edges_in_d=H, attention=True, edge_update=True)
    hr, xr, er = h, x, ea
    hg, xg, eg = h.to(dev), x.to(dev), ea.to(dev)
    rg, cg = row.to(dev), col.to(dev)
    for i in range(3):
        sd_np = synthetic_egcl_state_dict(H, H, 0, True, True, 40 + i, coord_gain=0.3)
        with torch.no_grad():
            hr, xr, er = orc.e_gcl_forward(orc.as_torch_sd(sd_np), cfg, hr, row, col, xr, er, nm, em)
        m = E_GCL(H, H, H, edges_in_d=H, attention=True, tanh=True, coords_range=30, edge_update=True)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
        m = m.to(dev)
        hg, xg, eg = m(hg, [rg, cg], xg, edge_attr=eg, node_mask=nm.to(dev), edge_mask=em.to(dev))
    assert_parity(hg.cpu().numpy(), hr.numpy(), "stack h")
    assert_parity(xg.cpu().numpy(), xr.numpy(), "stack x")
    assert_parity(eg.cpu().numpy(), er.numpy(), "stack edge_attr")


@pytest.mark.gpu
def test_randomised_layer_sweep():
    """tests/fuzz_stage2.py: random graphs / widths / options of the stage-2 layer against the oracle (a 300-case run of the same
    script: profiles/r03_fuzz_parity.log)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_stage2.py"), "40", "21"], cwd=root,
                          capture_output=True, text=True, timeout=600)
    tail = "\n".join(proc.stdout.splitlines()[-4:])
    assert proc.returncode == 0, tail + proc.stderr[-2000:]
    assert "failures 0" in tail, tail
