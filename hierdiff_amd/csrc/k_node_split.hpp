// The node update of the fp16x3 mode for FEW rows (round 5): k_node<..., F16>'s arithmetic, bit for bit, as three launches whose
// workgroups own a 32-row x 32-column output tile each.  Included through kernels.hpp.
//
// Why.  The fused kernel (k_node.hpp) gives a 32-row tile to ONE workgroup, which then streams 1.3 - 1.8 MB of weight images
// through one CU's vector-memory path (about 64 B / clock): 20 - 27 us however few tiles there are.  Below ~2,000 rows (B = 64 at
// N = 30; the reference's shipped job is B = 2) the other CUs idle meanwhile, and the node side is 37 - 62 % of a forward.  Here a
// phase's columns are spread over workgroups (64 KB of weights each instead of 0.5 MB), at the price of the two global
// synchronisations between the three dependent contractions, paid as kernel boundaries:
//     phase 1   T  = SiLU([h | agg] W3^T + b3)           grid: row tiles x H / 32
//     phase 2   h' = (h + T W4^T + b4) mask              grid: row tiles x H / 32
//     phase 3   AB_q = h' [W1a | W1b]_q^T + [b1 | 0]     grid: row tiles x 2 H / 32 x images
// A workgroup = four wavefronts = the four QUARTERS of the contraction's K range: wavefront w runs k-steps [w KS/4, (w+1) KS/4)
// into its own accumulator (its weight fragments - 8 or 16 KiB - are all requested at kernel entry), the four partial tiles meet in
// LDS and are added as ((q0 + q1) + q2) + q3.  k_node<..., F16> sums its contractions in exactly these quarters (NodeMma KQ = 4),
// ranges its operands by the same row bounds (which are functions of the row's max |h| and max |[h | agg]| only: phase 1 leaves
// the two numbers in `rowinfo` for the later phases) and splits the same fp32 values, so the two paths agree bit for bit and a
// sample's bits do not depend on which one its batch size selects (tests/test_gpu_parity.py::test_fp16x3_node_paths_agree_bitwise).
#pragma once
#include "common.hpp"
#include "k_node.hpp"

struct NodeSplitArgs {
    const float* h_in;      // [M_pad][H] node features (phase 1: operand; phase 2: the residual; phase 3, AB only: operand)
    const float* part;      // phase 1: [P][H] partial neighbour sums
    const int* pstart;      // phase 1: [M+1]
    const float* nmask;     // [M_pad]
    const float* Wimg[2];   // weight image(s) of the phase (pack_node_b_f16)
    const float* bias[2];
    float winv[2];          // 1 / (power-of-two scale of the image)
    float w3l1, w4l1, b3max, b4max;      // constants of the a-priori row bounds (k_node.hpp)
    float norm;
    float* T;               // phase 1 out, phase 2 in: [M_pad][H]
    float* h_out;           // phase 2 out, phase 3 in (may alias h_in: phase 2 touches its own 32 x 32 tile of it only)
    float* rowinfo;         // [M_pad][2] {max |h_r|, max |[h | agg]_r|}: written by phase 1, read by phases 2 and 3
    float* ABout[2];        // phase 3: [M_pad][2H]
    float* ABmax[2];        // phase 3, optional: [M_pad][2] row maxima of the two halves (atomic max; zeroed by phase 2 / the caller)
    float* zero_max[2];     // phase 2: the ABmax tables phase 3 is going to fill (zeroed here, one row tile per ct == 0 workgroup)
    int M, n_img, upd;
    // k_node_split_f32 as the node side of the stage-2 layer (hd_egcl_forward, round 5): phase 1's second operand half as a dense
    // [M][H] array of finished sums instead of partial sums; phase 2 without a residual (non-recurrent layer)
    const float* agg_dense;
    int resid_none;
};

// PH 1 / 2 / 3 as above.  H = 128 or 256.  CTW = 32-column tiles per workgroup; the library launches CTW = 1 (CTW = 2 - half the
// workgroups, half the redundant operand-tile loads, same bits - was measured slower from 24 to 64 molecules).
template <int H, int PH, int CTW>
__global__ __launch_bounds__(256, 2) void k_node_split(NodeSplitArgs a) {
    constexpr int K = PH == 1 ? 2 * H : H;
    constexpr int NCTW = (PH == 3 ? 2 * H : H) / 32;    // column tiles of the phase's output = of its weight image
    constexpr int KS = K / 16, KQ = KS / 4;             // k-steps, k-steps per wavefront
    static_assert(KS % 4 == 0, "four K quarters");
    constexpr int LD = K + 8;                           // fp16 row stride of an operand plane (16 B pad: conflict-free ds_read_b128)
    constexpr int PLANE = 32 * LD;
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    _Float16* Ah = reinterpret_cast<_Float16*>(smem_s);                              // [2 planes][32][LD]
    float* red = reinterpret_cast<float*>(smem_s);                                   // [4 quarters][CTW][16][64], over the operand planes once every wavefront is done with them (two workgroups per CU)
    __shared__ float un_s[32];                                                       // per row: 1 / (row scale x weight scale)
    __shared__ float sc_s[32];                                                       // phase 1: unused; kept for symmetry of the epilogue code

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    // block -> (row tile, column tile, image): every XCD owns a contiguous range of row tiles (k_node's split) and walks
    // (row tile, column tile) with the column tile fastest, so the 8 - 32 workgroups that read one operand tile share an L2
    int rt, ct, img = 0;
    {
        const int nrt = (a.M + 31) >> 5;
        const int per = NCTW / CTW * (PH == 3 ? a.n_img : 1);
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = nrt >> 3, r = nrt & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        if (idx >= len * per) return;
        rt = start + idx / per;
        const int rem = idx % per;
        if constexpr (PH == 3) { img = rem / (NCTW / CTW); ct = CTW * (rem % (NCTW / CTW)); } else ct = CTW * rem;
    }
    const int row0 = rt * 32;

    // this wavefront's weight fragments: all of its K quarter, requested before anything else (they do not depend on data)
    const u32x4* Wl = reinterpret_cast<const u32x4*>(a.Wimg[img]) + lane;
    u32x4 bf[KQ][CTW][2];
#pragma unroll
    for (int s = 0; s < KQ; ++s)
#pragma unroll
        for (int cc = 0; cc < CTW; ++cc)
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[s][cc][p] = Wl[((size_t)((wave * KQ + s) * NCTW + ct + cc) * 2 + p) * 64];
    float bias_v[CTW];
#pragma unroll
    for (int cc = 0; cc < CTW; ++cc) bias_v[cc] = a.bias[img][32 * (ct + cc) + n];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- operand tile -> LDS as fp16 head / tail, scaled per row (k_node phase 0 / its hand-overs, same expressions)
    {
        constexpr int Q = H / 4;                     // float4 per H-wide row
        constexpr int TPR = 8;                       // threads per row
        constexpr int NP = Q / TPR;                  // float4 per thread and H-wide source
        const int r = tid / TPR, cq = tid % TPR;
        const int row = row0 + r;
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        float sA = 1.0f;
        auto put = [&](int col, f32x4 v) {
            v = f32x4{v[0] * sA, v[1] * sA, v[2] * sA, v[3] * sA};
            const f16x4_t hi = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            const f16x4_t lo = {(_Float16)(v[0] - (float)hi[0]), (_Float16)(v[1] - (float)hi[1]), (_Float16)(v[2] - (float)hi[2]),
                                (_Float16)(v[3] - (float)hi[3])};
            *reinterpret_cast<f16x4_t*>(Ah + r * LD + col) = hi;
            *reinterpret_cast<f16x4_t*>(Ah + PLANE + r * LD + col) = lo;
        };
        if constexpr (PH == 1) {
            int p0 = 0, p1 = 0;
            if (row < a.M) { p0 = a.pstart[row]; p1 = a.pstart[row + 1]; }
            f32x4 hv[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) hv[u] = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)row * H + 4 * (cq + u * TPR));
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const bool has0 = p0 < p1, has1 = p0 + 1 < p1;
            const float* s0 = a.part + (size_t)(has0 ? p0 : 0) * H;
            const float* s1 = a.part + (size_t)(has1 ? p0 + 1 : 0) * H;
            f32x4 g0[NP], g1[NP], gv[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                g0[u] = *reinterpret_cast<const f32x4*>(s0 + 4 * (cq + u * TPR));
                g1[u] = *reinterpret_cast<const f32x4*>(s1 + 4 * (cq + u * TPR));
            }
            float mh = 0.f, mg = 0.f;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                f32x4 v = z4;
                if (has0) v += g0[u];
                if (has1) v += g1[u];
                for (int p = p0 + 2; p < p1; ++p) v += *reinterpret_cast<const f32x4*>(a.part + (size_t)p * H + 4 * (cq + u * TPR));
                gv[u] = v / a.norm;
#pragma unroll
                for (int j = 0; j < 4; ++j) { mh = fmaxf(mh, fabsf(hv[u][j])); mg = fmaxf(mg, fabsf(gv[u][j])); }
            }
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1) { mh = fmaxf(mh, __shfl_xor(mh, o)); mg = fmaxf(mg, __shfl_xor(mg, o)); }
            const float mx = fmaxf(mh, mg);
            float iX;
            f16_row_scale(mx, sA, iX);
            if (cq == 0) {
                un_s[r] = iX * a.winv[0];
                if (ct == 0 && row < a.M) { a.rowinfo[2 * (size_t)row] = mh; a.rowinfo[2 * (size_t)row + 1] = mx; }
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) put(4 * (cq + u * TPR), hv[u]);
#pragma unroll
            for (int u = 0; u < NP; ++u) put(H + 4 * (cq + u * TPR), gv[u]);
        } else {
            const float* src = PH == 2 ? a.T : (a.upd ? a.h_out : a.h_in);
            f32x4 v[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)row * H + 4 * (cq + u * TPR));     // pad rows are zero
            float inv;
            if (PH == 2 || a.upd) {
                const float mh = a.rowinfo[2 * (size_t)row], mx = a.rowinfo[2 * (size_t)row + 1];
                const float tb = __builtin_fmaf(mx, a.w3l1, a.b3max);
                const float bound = PH == 2 ? tb : mh + __builtin_fmaf(tb, a.w4l1, a.b4max);
                f16_row_scale(bound, sA, inv);
            } else {                                 // AB only (once per forward, behind the embedding): the row maximum of h itself
                float mh = 0.f;
#pragma unroll
                for (int u = 0; u < NP; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mh = fmaxf(mh, fabsf(v[u][j]));
#pragma unroll
                for (int o = TPR / 2; o > 0; o >>= 1) mh = fmaxf(mh, __shfl_xor(mh, o));
                f16_row_scale(mh, sA, inv);
            }
            if (cq == 0) {
                un_s[r] = inv * a.winv[img];
                if constexpr (PH == 2) {
                    if (ct == 0 && row < a.M) {      // the row maxima phase 3 accumulates with an atomic max start at zero
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            if (a.zero_max[q]) { a.zero_max[q][2 * (size_t)row] = 0.f; a.zero_max[q][2 * (size_t)row + 1] = 0.f; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) put(4 * (cq + u * TPR), v[u]);
        }
    }
    (void)sc_s;
    __syncthreads();

    // ---- this wavefront's K quarter: a_h b_h + a_l b_h + a_h b_l per k-step, on its own accumulator (from zero)
    {
        const _Float16* ap_h = Ah + n * LD + 8 * hh + 16 * (wave * KQ);
        const _Float16* ap_l = ap_h + PLANE;
        f32x16 acc[CTW];
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap_h + 16 * s), al = *reinterpret_cast<const f16x8*>(ap_l + 16 * s);
#pragma unroll
            for (int cc = 0; cc < CTW; ++cc) {
                const f16x8 bh = __builtin_bit_cast(f16x8, bf[s][cc][0]), bl = __builtin_bit_cast(f16x8, bf[s][cc][1]);
                if (s == 0) acc[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, f32x16{}, 0, 0, 0);
                else acc[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[cc], 0, 0, 0);
                acc[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[cc], 0, 0, 0);
                acc[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[cc], 0, 0, 0);
            }
        }
        __syncthreads();                 // every wavefront has read its operand rows: the planes become the exchange buffer
#pragma unroll
        for (int cc = 0; cc < CTW; ++cc)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * CTW + cc) * 16 + r) * 64 + lane] = acc[cc][r];
    }
    __syncthreads();

    // ---- the four quarters meet: wavefront w finishes accumulator rows r = 4w .. 4w + 3 of every lane
#pragma unroll
    for (int cc = 0; cc < CTW; ++cc)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * wave + j;
        const int R = (r & 3) + 8 * (r >> 2) + 4 * hh;            // tile row of accumulator element r in lane half hh
        const int row = row0 + R, col = 32 * (ct + cc) + n;
        auto part_of = [&](int q) { return red[((q * CTW + cc) * 16 + r) * 64 + lane]; };
        const float v = ((part_of(0) + part_of(1)) + part_of(2)) + part_of(3);
        const float x = __builtin_fmaf(v, un_s[R], bias_v[cc]);
        if constexpr (PH == 1) {
            if (row < a.M) a.T[(size_t)row * H + col] = silu_f(x);
        } else if constexpr (PH == 2) {
            if (row < a.M) {
                const float hres = a.h_in[(size_t)row * H + col];
                a.h_out[(size_t)row * H + col] = (hres + x) * a.nmask[row];
            }
        } else {
            const int half = (ct + cc) / (H / 32), c = col - half * H;
            if (row < a.M) a.ABout[img][(size_t)row * 2 * H + half * H + c] = x;
            if (a.ABmax[img]) {
                float m = fabsf(x);
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                // non-negative floats order like their bit patterns; a NaN (sign bit clear or not) is handed on as the largest pattern
                if (n == 0 && row < a.M) atomicMax(reinterpret_cast<unsigned*>(a.ABmax[img] + 2 * (size_t)row + half), __builtin_bit_cast(unsigned, m));
            }
        }
    }
}

template <int H, int PH, int CTW>
constexpr int node_split_lds_bytes() {
    const int planes = 2 * 32 * ((PH == 1 ? 2 * H : H) + 8) * 2, red = 4 * CTW * 16 * 64 * 4;
    return planes > red ? planes : red;
}


// ----------------------------------------------------------------------------- the same chain in exact fp32 (widths >= 128)
// k_node_f32's arithmetic, bit for bit (it sums its contractions in the same four K quarters, NodeMmaF NQ = 4), for batches below
// HD_FUSE_MIN_ROWS active rows: 32 x 32 output tiles over many workgroups, a quarter of the K range per wavefront
// (v_mfma_f32_32x32x2_f32: 64 dependent instructions for K = 512 instead of the 128 of k_gemm_r16's 16 x 16 x 4 chain and the
// 256 of an unsplit 32 x 32 x 2 one), all of a wavefront's weight fragments (k_node_f32's images: [32-wide K chunk][column tile]
// [4 q][64 lanes][4 j]) requested at entry.  Narrow widths keep k_gemm_r16 (their fused kernel sums in one chain).
template <int H, int PH>
__global__ __launch_bounds__(256, 2) void k_node_split_f32(NodeSplitArgs a) {
    constexpr int K = PH == 1 ? 2 * H : H;
    constexpr int NCTW = (PH == 3 ? 2 * H : H) / 32;
    constexpr int KC = K / 32, KQ = KC / 4;             // 32-wide K chunks, chunks per wavefront
    static_assert(KC % 4 == 0, "four K quarters");
    constexpr int LD = K + 4;                           // fp32 row stride (conflict-free ds_read_b128, as in k_node_f32)
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    float* X = reinterpret_cast<float*>(smem_s);        // [32][LD]
    float* red = reinterpret_cast<float*>(smem_s);      // [4 quarters][16][64], over the operand tile once every wavefront is done with it

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    int rt, ct, img = 0;
    {
        const int nrt = (a.M + 31) >> 5;
        const int per = NCTW * (PH == 3 ? a.n_img : 1);
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = nrt >> 3, r = nrt & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        if (idx >= len * per) return;
        rt = start + idx / per;
        const int rem = idx % per;
        if constexpr (PH == 3) { img = rem / NCTW; ct = rem % NCTW; } else ct = rem;
    }
    const int row0 = rt * 32;

    const u32x4* Wl = reinterpret_cast<const u32x4*>(a.Wimg[img]) + lane;
    u32x4 bf[KQ][4];
#pragma unroll
    for (int s = 0; s < KQ; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) bf[s][q] = Wl[((size_t)((wave * KQ + s) * NCTW + ct) * 4 + q) * 64];
    const float bias_v = a.bias[img][32 * ct + n];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- operand tile -> LDS (k_node_f32 phase 0 / its hand-overs, same expressions)
    {
        constexpr int Q = H / 4, TPR = 8, NP = Q / TPR;
        const int r = tid / TPR, cq = tid % TPR;
        const int row = row0 + r;
        const int lrow = row < a.M ? row : a.M - 1;      // rows past the end are never stored: read a valid one (the stage-2 layer hands unpadded tensors)
        if constexpr (PH == 1) {
            int p0 = 0, p1 = 0;
            if (row < a.M && !a.agg_dense) { p0 = a.pstart[row]; p1 = a.pstart[row + 1]; }
            f32x4 hv[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) hv[u] = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)lrow * H + 4 * (cq + u * TPR));
            if (a.agg_dense) {
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    *reinterpret_cast<f32x4*>(X + r * LD + 4 * (cq + u * TPR)) = hv[u];
                    *reinterpret_cast<f32x4*>(X + r * LD + H + 4 * (cq + u * TPR)) =
                        *reinterpret_cast<const f32x4*>(a.agg_dense + (size_t)lrow * H + 4 * (cq + u * TPR));
                }
            } else {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const bool has0 = p0 < p1, has1 = p0 + 1 < p1;
            const float* s0 = a.part + (size_t)(has0 ? p0 : 0) * H;
            const float* s1 = a.part + (size_t)(has1 ? p0 + 1 : 0) * H;
            f32x4 g0[NP], g1[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                g0[u] = *reinterpret_cast<const f32x4*>(s0 + 4 * (cq + u * TPR));
                g1[u] = *reinterpret_cast<const f32x4*>(s1 + 4 * (cq + u * TPR));
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) *reinterpret_cast<f32x4*>(X + r * LD + 4 * (cq + u * TPR)) = hv[u];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                f32x4 v = z4;                        // parts ascending, then / norm (k_node_f32, k_agg)
                if (has0) v += g0[u];
                if (has1) v += g1[u];
                for (int p = p0 + 2; p < p1; ++p) v += *reinterpret_cast<const f32x4*>(a.part + (size_t)p * H + 4 * (cq + u * TPR));
                *reinterpret_cast<f32x4*>(X + r * LD + H + 4 * (cq + u * TPR)) = v / a.norm;
            }
            }
        } else {
            const float* src = PH == 2 ? a.T : (a.upd ? a.h_out : a.h_in);
#pragma unroll
            for (int u = 0; u < NP; ++u)
                *reinterpret_cast<f32x4*>(X + r * LD + 4 * (cq + u * TPR)) = *reinterpret_cast<const f32x4*>(src + (size_t)lrow * H + 4 * (cq + u * TPR));
        }
    }
    __syncthreads();

    // ---- this wavefront's K quarter on its own accumulator (from zero): chunks ascending, q, j - k_node_f32's order
    {
        const float* Arow = X + n * LD + 16 * hh + 32 * (wave * KQ);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KQ; ++s) {
            f32x4 av[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(Arow + 32 * s + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = __builtin_bit_cast(f32x4, bf[s][q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][j], b[j], acc, 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * wave + j;
        const int R = (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int row = row0 + R, col = 32 * ct + n;
        const float v = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) + red[(3 * 16 + r) * 64 + lane];
        if (row >= a.M) continue;
        if constexpr (PH == 1) a.T[(size_t)row * H + col] = silu_f(v + bias_v);
        else if constexpr (PH == 2) {
            if (a.resid_none) a.h_out[(size_t)row * H + col] = (v + bias_v) * a.nmask[row];
            else a.h_out[(size_t)row * H + col] = (a.h_in[(size_t)row * H + col] + (v + bias_v)) * a.nmask[row];
        }
        else a.ABout[img][(size_t)row * 2 * H + col] = v + bias_v;
    }
}

template <int H, int PH>
constexpr int node_split_f32_lds_bytes() {
    const int tile = 32 * ((PH == 1 ? 2 * H : H) + 4) * 4, red = 4 * 16 * 64 * 4;
    return tile > red ? tile : red;
}
