// The variational training loss around the network call, fused (round 5): compute_loss of the reference in training mode
// (diffusion_qm9.py:530-673 with t0_always = False; compute_error :160-172, kl_prior :206-239, log_constants :241-262,
// log_pxh_given_z0_without_constants :460-528) as ONE kernel per direction instead of ~350 element-wise launches of [B, N, 11]
// tensors - at the reference's own batch size (conf/dataset/geom_blur.yaml: 16) a training step is bound by the number of launches,
// not by what they compute.  Included through kernels.hpp.  One workgroup per molecule; everything is fp32 like the torch ops it
// replaces (same expressions, sums in another order: results agree to round-off, tests/test_gpu_training.py).
//
//   z_t   = alpha_t xh + sigma_t eps,  alpha_t = sqrt(sigmoid(-g_t)), sigma_t = sqrt(sigmoid(g_t))        (k_vlb_zt, k_vlb_zt_bwd)
//   E     = sum (eps - net)^2 [x cE]                        cE = 1, or 1 / ((3 + F) N) for the `l2` training loss
//   Lpos  = 1/2 w E,  w = exp(g_t - g_s) - 1 (vlb) | 1 (l2)
//   L0    = 1/2 cE (Ex + Eh) - sum nm log(Phi((c + 1/2) / s0) - Phi((c - 1/2) / s0) + 1e-10)                the t = 0 term: Ex over the
//           coordinate columns, Eh = sum_n sum_{c < cont} (eps[n][3 + int + c] - net[n][0])^2 (the reference's strided slice),
//           c = round(h nv2 + nb2) - (z_t nv2 + nb2) over the `int` integer feature columns, s0 = sigma_t nv2
//   K     = (n F + d)(-log sigma_T + 1/2 sigma_T^2 - 1/2) + 1/2 alpha_T^2 sum xh^2,  d = 3 (n - 1)         KL to the prior
//   C0    = (d + n F)(1/2 g_0 + 1/2 log 2 pi)   (vlb)  |  0 (l2)
//   loss  = K + (T + 1 | 1)(z L0 + (1 - z) Lpos) + C0 - delta,   z = [t_int == 0],  delta = -d log nv0 (vlb) | 0 (l2)
// Backward: d loss / d net, d loss / d z_t (only through L0's integer likelihood), d loss / d (g_s, g_t, g_0, g_T) - the schedule
// network is trained (conf/model/ddpmgblur.yaml: noise_schedule learned).
#pragma once
#include "common.hpp"

struct VlbArgs {
    const float* net;       // [B][N][D] network output (eps prediction)
    const float* zt;        // [B][N][D]
    const float* xh;        // [B][N][D] normalised data [x | h]
    const float* eps;       // [B][N][D]
    const float* nm;        // [B][N] node mask as floats
    const float* gam;       // [4][B]: g_s, g_t, g_0, g_T
    const float* t_int;     // [B]
    float* loss;            // [B]
    float* err;             // [B] E (the `error` entry of the reference's info dict)
    // backward
    const float* gout;      // [B] d L / d loss
    float* dnet;            // [B][N][D]
    float* dzt;             // [B][N][D]
    float* dgam;            // [4][B]
    int B, N, D, int_nf, cont_nf, l2_train;
    float T, nv2, nb2, log_nv0;
};

HD_DEVINL float vlb_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
HD_DEVINL float vlb_cdf(float v) { return 0.5f * (1.0f + erff(v * 0.70710678118654752f)); }
HD_DEVINL float vlb_pdf(float v) { return 0.39894228040143268f * expf(-0.5f * v * v); }

// sum of `NV` per-thread values over a 256-thread workgroup; the result is valid in every thread
template <int NV>
HD_DEVINL void vlb_block_sum(float (&v)[NV], float* red /* [4][NV] */) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = (red[k] + red[NV + k]) + (red[2 * NV + k] + red[3 * NV + k]);
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_vlb(VlbArgs a) {
    __shared__ float red[4 * 8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = a.N, D = a.D, nd = 3;
    const size_t base = (size_t)b * N * D;
    const float gs = a.gam[b], gt = a.gam[a.B + b], g0 = a.gam[2 * a.B + b], gT = a.gam[3 * a.B + b];
    const float sig_t = vlb_sigmoid(gt);
    const float sigma_t = sqrtf(sig_t);
    const float s0 = sigma_t * a.nv2, inv_s0 = 1.0f / s0;
    const float cE = a.l2_train ? 1.0f / (float)((nd + (D - nd)) * N) : 1.0f;
    const bool t0 = a.t_int[b] == 0.0f;
    // pass 1: the sums.  v = {E, Ex + Eh, sum xh^2 (h masked), n, log_int, d log_int / d s0, -, -}
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < N * D; idx += 256) {
        const int n = idx / D, d = idx - n * D;
        const float m = a.nm[(size_t)b * N + n];
        const float e = a.eps[base + idx], o = a.net[base + idx], x = a.xh[base + idx];
        const float r = e - o;
        v[0] += r * r;
        if (d < nd) v[1] += r * r;
        v[2] += (d < nd) ? x * x : m * x * x;
        if (d == 0) v[3] += m;
        if (d >= nd + a.int_nf && d < nd + a.int_nf + a.cont_nf) {                 // eps[n][3 + int + c] against net[n][0]
            const float rc = e - a.net[base + (size_t)n * D];
            v[1] += rc * rc;
        }
        if (d >= nd && d < nd + a.int_nf) {
            const float hint = rintf(x * a.nv2 + a.nb2);
            const float c = hint - (a.zt[base + idx] * a.nv2 + a.nb2);
            const float ap = (c + 0.5f) * inv_s0, am = (c - 0.5f) * inv_s0;
            const float dphi = vlb_cdf(ap) - vlb_cdf(am) + 1e-10f;
            v[4] += m * logf(dphi);
            if (BWD) v[5] += -m * (ap * vlb_pdf(ap) - am * vlb_pdf(am)) * inv_s0 / dphi;
        }
    }
    vlb_block_sum<6>(v, red);
    const float E = v[0] * cE, E0 = v[1] * cE, S = v[2], n = v[3], logint = v[4];
    const float F = (float)(D - nd), dsub = (n - 1.0f) * nd;
    const float w = a.l2_train ? 1.0f : expf(gt - gs) - 1.0f;
    const float Lpos = 0.5f * w * E;
    const float L0 = 0.5f * E0 - logint;
    const float uT = vlb_sigmoid(gT), aT2 = vlb_sigmoid(-gT);      // sigma_T^2, alpha_T^2 (= 1 - uT, without the cancellation)
    const float K = (n * F + dsub) * (-0.5f * logf(uT) + 0.5f * uT - 0.5f) + 0.5f * aT2 * S;
    const float C0 = a.l2_train ? 0.0f : (dsub + n * F) * (0.5f * g0 + 0.91893853320467274f);
    const float est = a.l2_train ? 1.0f : a.T + 1.0f;
    const float delta = a.l2_train ? 0.0f : -dsub * a.log_nv0;
    if (!BWD) {
        if (tid == 0) {
            a.loss[b] = K + est * (t0 ? L0 : Lpos) + C0 - delta;
            a.err[b] = E;
        }
        return;
    }
    // pass 2: the gradients
    const float go = a.gout[b];
    const float cnet = go * est * cE * (t0 ? 1.0f : w);     // d / d net of 1/2 cE (...)^2 terms: (net - eps) x this  (t = 0: x columns only)
    for (int idx = tid; idx < N * D; idx += 256) {
        const int nn = idx / D, d = idx - nn * D;
        const float e = a.eps[base + idx], o = a.net[base + idx];
        float g = 0.0f, gz = 0.0f;
        if (!t0) g = cnet * (o - e);
        else {
            if (d < nd) g = cnet * (o - e);
            if (d == 0) {
                for (int c = 0; c < a.cont_nf; ++c) g += cnet * (o - a.eps[base + (size_t)nn * D + nd + a.int_nf + c]);
            }
            if (d >= nd && d < nd + a.int_nf) {
                const float m = a.nm[(size_t)b * N + nn];
                const float hint = rintf(a.xh[base + idx] * a.nv2 + a.nb2);
                const float c = hint - (a.zt[base + idx] * a.nv2 + a.nb2);
                const float ap = (c + 0.5f) * inv_s0, am = (c - 0.5f) * inv_s0;
                const float dphi = vlb_cdf(ap) - vlb_cdf(am) + 1e-10f;
                // L0 = ... - log_int;  d log_int / d z = m (pdf(ap) - pdf(am)) / dphi x (-nv2 / s0)
                gz = go * est * m * (vlb_pdf(ap) - vlb_pdf(am)) / dphi * (a.nv2 * inv_s0);
            }
        }
        a.dnet[base + idx] = g;
        a.dzt[base + idx] = gz;
    }
    if (tid == 0) {
        float dgs = 0.f, dgt = 0.f, dg0 = 0.f, dgT = 0.f;
        if (!a.l2_train) dg0 = 0.5f * (dsub + n * F);
        // d K / d g_T = 1/2 u (1 - u) [(n F + d)(1 - 1/u) - S] with 1 - 1/u = -exp(-g_T): g_T ~ 10, where 1 - 1/u in fp32 is noise
        dgT = 0.5f * uT * aT2 * (-(n * F + dsub) * expf(-gT) - S);
        if (!t0) {
            if (!a.l2_train) { const float ex = expf(gt - gs); dgt = est * 0.5f * E * ex; dgs = -est * 0.5f * E * ex; }
        } else {
            // L0 depends on g_t through s0 = sigma_t nv2:  d sigma_t / d g_t = 1/2 sigma_t (1 - sigmoid(g_t))
            dgt = est * (-v[5]) * a.nv2 * 0.5f * sigma_t * vlb_sigmoid(-gt);
        }
        a.dgam[b] = go * dgs; a.dgam[a.B + b] = go * dgt; a.dgam[2 * a.B + b] = go * dg0; a.dgam[3 * a.B + b] = go * dgT;
    }
}

// z_t = alpha_t xh + sigma_t eps  /  its backward with respect to g_t: dg[b] = sum dz (xh d alpha / d g + eps d sigma / d g)
struct VlbZtArgs { const float* xh; const float* eps; const float* gt; float* zt; const float* dzt; float* dgt; int B, ND; };

template <bool BWD>
__global__ __launch_bounds__(256) void k_vlb_zt(VlbZtArgs a) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)b * a.ND;
    const float g = a.gt[b];
    const float sp = vlb_sigmoid(g), sn = vlb_sigmoid(-g);
    const float sigma = sqrtf(sp), alpha = sqrtf(sn);
    if (!BWD) {
        for (int i = tid; i < a.ND; i += 256) a.zt[base + i] = alpha * a.xh[base + i] + sigma * a.eps[base + i];
        return;
    }
    const float dsig = 0.5f * sigma * sn, dalp = -0.5f * alpha * sp;
    float v[1] = {0.f};
    for (int i = tid; i < a.ND; i += 256) v[0] += a.dzt[base + i] * (a.xh[base + i] * dalp + a.eps[base + i] * dsig);
    vlb_block_sum<1>(v, red);
    if (tid == 0) a.dgt[b] = v[0];
}

// ----------------------------------------------------------------------------- small layout kernels of the training path (launch count)
// The first edge Linear W1 [H][2H + 2] (columns: h_row | h_col | radial | d0) as the operands the edge layer wants, in ONE launch
// instead of torch's cat / cat / transpose: Wst [2H][H] = [W1[:, :H] ; W1[:, H:2H]], bst [2H] = [b1 | 0], wrd [2][H] = the two distance
// columns.  DIR 1 is the way back: dW1 [H][2H + 2] from dWst [2H][H] and dwrd [2][H].
struct EdgePrepArgs { const float* W1; const float* b1; float* Wst; float* bst; float* wrd; int H; };
template <int DIR>
__global__ __launch_bounds__(256) void k_edge_prep(EdgePrepArgs a) {
    const int H = a.H, ld = 2 * H + 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * ld) return;
    const int r = idx / ld, c = idx - r * ld;                 // element (r, c) of W1 / dW1
    float* w1 = const_cast<float*>(a.W1);
    if (c < 2 * H) {
        float* st = a.Wst + ((size_t)(c < H ? r : H + r)) * H + (c < H ? c : c - H);
        if (DIR == 0) *st = w1[idx]; else w1[idx] = *st;
    } else {
        float* wd = a.wrd + (size_t)(c - 2 * H) * H + r;
        if (DIR == 0) *wd = w1[idx]; else w1[idx] = *wd;
    }
    if (DIR == 0 && idx < 2 * H) a.bst[idx] = idx < H ? a.b1[idx] : 0.0f;
}
