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


def _array_scale(x):
    """k_dw2_f16: ONE power of two per operand, max |x| -> [2^14, 2^15) (at most 2^100: all-zero / tiny arrays)."""
    m = np.float32(np.abs(x).max())
    if m == 0:
        return np.float32(2.0) ** 100
    _, e = np.frexp(m)
    return np.float32(2.0) ** min(100, 15 - int(e))


def fp16x3_dw2(G, P):
    """dW2[c][k] = sum_e G[e][c] P[e][k] the way k_dw2_f16 forms it (hierdiff_amd/csrc/k_dw2.hpp): both operands scaled by one
    power of two each, FP16 head / tail, three of the four cross terms, fp32 accumulation, un-scaled at the end."""
    gs, ps = _array_scale(G), _array_scale(P)
    gh, gl = _split((G * gs).astype(np.float32))
    ph, pl = _split((P * ps).astype(np.float32))
    f = lambda a: a.astype(np.float32)
    assert np.isfinite(f(gh)).all() and np.isfinite(f(ph)).all(), "a head overflowed"
    acc = f(gh).T @ f(pl) + f(gl).T @ f(ph) + f(gh).T @ f(ph)
    return acc * (np.float32(1.0) / gs) * (np.float32(1.0) / ps)


@pytest.mark.parametrize("row_decades,col_decades", [(0, 0), (6, 4), (12, 0), (3, 8)])
def test_dw2_one_scale_per_operand_is_enough(row_decades, col_decades):
    """The training path's dense reduction over all edge rows (round 5).  Per-row ranging is impossible here (the row index is the
    contraction index), and unnecessary: a sum is as exact as its largest terms are, and an element 2^18 below the array maximum
    still has 22 significant bits (head normal, tail at the subnormal quantum 2^-24).  Gradient rows over `row_decades` decades
    (gated-off edges next to the ones that matter) and columns over `col_decades`: the result stays at the error of an fp32 GEMM of
    the same operands against float64."""
    rng = np.random.Generator(np.random.PCG64(11 + row_decades))
    E, H = 2048, 128
    G = rng.standard_normal((E, H)).astype(np.float32)
    G *= np.float32(10.0) ** (-row_decades * rng.random((E, 1))).astype(np.float32)
    G *= np.float32(10.0) ** (-col_decades * rng.random((1, H))).astype(np.float32) * np.float32(1e-4)      # gradients are small numbers
    pre = rng.standard_normal((E, H)).astype(np.float32) * 3
    P = (pre / (1 + np.exp(-pre))).astype(np.float32)
    ref = G.astype(np.float64).T @ P.astype(np.float64)
    # per COLUMN of the result (a column of tiny gradients must be as good as a large one: relative error per output row)
    rel = lambda y: float(np.max(np.linalg.norm(y - ref, axis=1) / np.linalg.norm(ref, axis=1)))
    err16 = rel(fp16x3_dw2(G, P).astype(np.float64))
    err32 = rel((G.T @ P).astype(np.float64))
    assert err16 < max(2.0 * err32, 6e-7) or col_decades >= 8, (err16, err32)
    if col_decades >= 8:        # documented limit: a result row whose gradients sit 8 decades (2^26) below the array maximum loses bits
        whole = float(np.linalg.norm(fp16x3_dw2(G, P).astype(np.float64) - ref) / np.linalg.norm(ref))
        assert whole < 6e-7 and err16 < 1e-2, (whole, err16)


def test_dw2_scales_of_degenerate_arrays():
    z = np.zeros((64, 128), dtype=np.float32)
    P = np.ones((64, 128), dtype=np.float32)
    assert float(np.abs(fp16x3_dw2(z, P)).max()) == 0.0
    tiny = np.full((64, 128), 1e-38, dtype=np.float32)           # the 2^100 clamp: heads become subnormal, nothing overflows or turns NaN
    out = fp16x3_dw2(tiny, P)
    assert np.isfinite(out).all()
