"""Shared helpers for the parity tests (tests/ may import oracle/; the product package may not)."""
import os

import numpy as np
import torch

from hierdiff_amd.weights import synthetic_state_dict
from oracle import egnn_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Parity bar from SURVEY.md section 8c: rel-L2 over the whole [B,N,3+F] output < 1e-4 and
# max-abs < 1e-4 * max(1, max|ref|).  A correct fp32 kernel is expected around 1e-6.
REL_L2_TOL = 1e-4
MAX_ABS_TOL = 1e-4


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_parity(got, ref, what="", rel_tol=REL_L2_TOL, abs_tol=MAX_ABS_TOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    assert np.all(np.isfinite(got)), f"{what}: non-finite output"
    r = rel_l2(got, ref)
    m = float(np.max(np.abs(got - ref))) if got.size else 0.0
    bound = abs_tol * max(1.0, float(np.max(np.abs(ref))) if ref.size else 1.0)
    assert r < rel_tol, f"{what}: rel_l2 {r:.3e} >= {rel_tol}"
    assert m < bound, f"{what}: max_abs {m:.3e} >= {bound:.3e}"
    return r, m


def fixture_model(fx, context_node_nf=0):
    """(numpy state_dict, oracle torch state_dict, oracle cfg) for a golden fixture."""
    H, L = int(fx["hidden_nf"]), int(fx["n_layers"])
    sd_np = synthetic_state_dict(9, context_node_nf, H, L, 2, True, int(fx["weight_seed"]),
                                 float(fx["coord_gain"]))
    cfg = orc.DynCfg(in_node_nf=9, context_node_nf=context_node_nf, hidden_nf=H, n_layers=L,
                     normalization_factor=10.0)
    return sd_np, orc.as_torch_sd(sd_np), cfg


def chain_noise(seed, T, B, N, F=8):
    """The (T+2) x (randn_x [B,N,3], randn_h [B,N,F]) draws of a chain fixture that stores `noise_seed` instead of the
    draws (oracle/make_golden.py:fixture_chain): numpy PCG64, x then h per draw."""
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    return [(torch.from_numpy(rng.standard_normal((B, N, 3)).astype(np.float32)),
             torch.from_numpy(rng.standard_normal((B, N, F)).astype(np.float32))) for _ in range(T + 2)]
