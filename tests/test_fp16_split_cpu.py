"""CPU restatement of the arithmetic of the `fp16x3` mode (hierdiff_amd/csrc/k_edge.hpp, PREC 3) in numpy: operands ranged by exact
powers of two, split into an FP16 head and tail (subnormal tails kept, as gfx950's matrix core does), three of the four cross
terms accumulated in fp32.  What the GPU tests measure on the real instruction is checked here as arithmetic: the ranging can never
overflow, and the truncation stays at the level of the fp32 accumulation whatever the magnitude of weights and activations."""
import numpy as np
import pytest


def _split(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def _weight_scale(W):
    """pack_edge_w2_f16: the power of two that puts max |W| into [2^14, 2^15)."""
    m, e = np.frexp(np.float32(np.abs(W).max()))          # max = m 2^e, m in [0.5, 1)
    return np.float32(2.0) ** (15 - int(e))


def _row_scale(bound):
    """k_edge.hpp: s = 2^(13 - E), E = floor(log2 bound)  =>  bound * s in [2^13, 2^14)."""
    E = np.floor(np.log2(bound.astype(np.float64))).astype(np.int64)
    return np.ldexp(np.float32(1.0), (13 - E).astype(np.int32)).astype(np.float32)


def fp16x3_contract(P, W, bound):
    """sum_k P[e][k] W[c][k] the way PREC 3 forms it; `bound[e]` >= max_k |P[e][k]|."""
    sw = _weight_scale(W)
    s = _row_scale(bound)[:, None]
    ph, pl = _split((P * s).astype(np.float32))
    wh, wl = _split((W * sw).astype(np.float32))
    assert np.isfinite(ph.astype(np.float32)).all() and np.isfinite(wh.astype(np.float32)).all(), "a head overflowed"
    f = lambda a: a.astype(np.float32)
    acc = f(ph) @ f(wh).T + f(pl) @ f(wh).T + f(ph) @ f(wl).T          # fp32 accumulation of exact fp16 x fp16 products
    return acc / (s * sw)


@pytest.mark.parametrize("p_gain,w_gain", [(1.0, 1.0), (1e-4, 1.0), (3e4, 1.0), (1.0, 1e-3), (1.0, 3e2), (1e6, 1e-5), (1e-6, 1e4)])
def test_fp16x3_arithmetic_is_fp32_accurate_at_any_magnitude(p_gain, w_gain):
    rng = np.random.Generator(np.random.PCG64(5))
    E, H = 512, 256
    pre = rng.standard_normal((E, H)).astype(np.float32) * 2
    P = (pre / (1 + np.exp(-pre))).astype(np.float32) * np.float32(p_gain)
    # rows of very different size in one tile, as edges with near and far endpoints are
    P *= np.float32(2.0) ** rng.integers(-6, 7, (E, 1)).astype(np.float32)
    W = ((rng.random((H, H)) * 2 - 1) / 16).astype(np.float32) * np.float32(w_gain)
    bound = np.abs(P).max(1) * np.float32(1.0 + 3.0 * rng.random(E))       # a valid, loose bound per row
    ref = P.astype(np.float64) @ W.astype(np.float64).T
    rel = lambda y: float(np.linalg.norm(y - ref) / np.linalg.norm(ref))
    err16 = rel(fp16x3_contract(P, W, bound).astype(np.float64))
    err32 = rel((P @ W.T).astype(np.float64))                               # plain fp32 GEMM of the same operands
    assert err16 < max(2.0 * err32, 6e-7), (err16, err32)


def test_fp16x3_ranging_cannot_overflow():
    """bound * s < 2^14 for every positive finite bound, so |P| <= bound keeps the head below 65504; the weight image peaks
    below 2^15."""
    bounds = np.float32(2.0) ** np.arange(-60, 61, dtype=np.float32) * np.float32(1.999)
    s = _row_scale(bounds)
    assert ((bounds * s) < 2.0 ** 14).all() and ((bounds * s) >= 2.0 ** 13).all()
    for wmax in (1e-12, 3e-3, 0.06, 1.0, 17.0, 9e8):
        W = np.array([[wmax, -wmax / 3]], dtype=np.float32)
        assert 2.0 ** 14 <= float(np.abs(W).max() * _weight_scale(W)) < 2.0 ** 15


def test_node_update_row_bounds_hold_and_are_usable():
    """k_node<..., F16> ranges T and h' per row from bounds known before the first contraction (k_node.hpp):
    |T_r| <= max|X_r| max_c sum_k|W3[c][k]| + max|b3|,  |h'_r| <= max|h_r| + bound(T_r) max_c sum_k|W4[c][k]| + max|b4|.
    They must hold for every element (no FP16 overflow possible) and must not be so loose that typical elements fall out of the
    22-bit range of a scaled operand (15 binades below the bound)."""
    rng = np.random.Generator(np.random.PCG64(9))
    H, R = 256, 200
    for gain_h, gain_w in ((1.0, 1.0), (300.0, 1.0), (1e-3, 1.0), (1.0, 40.0), (1.0, 1e-2)):
        h = rng.standard_normal((R, H)).astype(np.float32) * np.float32(gain_h) * (2.0 ** rng.integers(-4, 5, (R, 1))).astype(np.float32)
        agg = rng.standard_normal((R, H)).astype(np.float32) * np.float32(gain_h)
        X = np.concatenate([h, agg], 1)
        W3 = ((rng.random((H, 2 * H)) * 2 - 1) / 22).astype(np.float32) * np.float32(gain_w)
        W4 = ((rng.random((H, H)) * 2 - 1) / 16).astype(np.float32) * np.float32(gain_w)
        b3 = rng.standard_normal(H).astype(np.float32) * 0.1
        b4 = rng.standard_normal(H).astype(np.float32) * 0.1
        pre = X.astype(np.float64) @ W3.astype(np.float64).T + b3
        T = pre / (1 + np.exp(-np.clip(pre, -700, 700)))
        hn = h + T @ W4.astype(np.float64).T + b4
        mx, mh = np.abs(X).max(1), np.abs(h).max(1)
        tb = mx * np.abs(W3).sum(1).max() + np.abs(b3).max()
        hb = mh + tb * np.abs(W4).sum(1).max() + np.abs(b4).max()
        assert (np.abs(T).max(1) <= tb * (1 + 1e-6)).all() and (np.abs(hn).max(1) <= hb * (1 + 1e-6)).all()
        # looseness: the row's own largest element sits fewer than 12 binades below its bound, i.e. at least 3 binades inside
        # the range where an element keeps both FP16 pieces normal
        assert (np.log2(tb / np.maximum(np.abs(T).max(1), 1e-300)) < 12).all()
        assert (np.log2(hb / np.maximum(np.abs(hn).max(1), 1e-300)) < 12).all()
