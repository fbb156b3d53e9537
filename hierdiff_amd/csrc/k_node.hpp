// Node-side kernels: input embedding (k_node_init), the fp32 tile GEMM k_gemm (stage-2 layer E_GCL; the sampler's fp32 node
// chain of small batches is k_gemm_r16.hpp), and the fused node update
// of the sampler in its three arithmetics (k_node: bf16x3 / bf16x6; k_node_f32: exact fp32).  Included through kernels.hpp.
#pragma once
#include "common.hpp"

// ----------------------------------------------------------------------------- node init
// xh*mask -> x0/xcur; [h*mask | t | context] -> embedding (en_dynamics.py:57-79, egnn_new.py:197).

struct InitArgs {
    const float* xh;        // [B*N][D]
    const float* t;         // [1] or [B]
    const float* ctx;       // [B*N][C] or null
    const int* node_of;     // [M] compact -> flat
    const float* nmask;     // [M_pad] 0/1
    const float* embT;      // [fin][H]
    const float* emb_b;     // [H]
    float* h;               // [M_pad][H]
    float* x0;              // [M_pad][4]
    float* xcur;            // [M_pad][4]
    int* nanflag;           // reset here (first kernel of a forward), raised by k_post1, consumed by k_post2
    float* zero_max;        // optional [M_pad][2]: row maxima the split node chain (k_node_split.hpp) accumulates with an atomic max
    int M, N, D, F, C, H, t_stride, cond_time;
};

// One thread per (node, 4 consecutive output columns): the node's feature row is read once per thread instead of
// once per output element, weights and the result move as float4.
__global__ void k_node_init(InitArgs a) {
    const int q = a.H >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) *a.nanflag = 0;
    const int i = idx / q, c = 4 * (idx - i * q);
    if (i >= a.M) return;
    const int flat = a.node_of[i];
    const float m = a.nmask[i];
    const float* row = a.xh + (size_t)flat * a.D;
    f32x4 acc = *reinterpret_cast<const f32x4*>(a.emb_b + c);
    auto fma4 = [&](float v, int f) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.embT + (size_t)f * a.H + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(v, w[j], acc[j]);
    };
    int f = 0;
    for (; f < a.F; ++f) fma4(row[3 + f] * m, f);
    if (a.cond_time) { fma4(a.t[(flat / a.N) * a.t_stride], f); ++f; }
    for (int k = 0; k < a.C; ++k, ++f) fma4(a.ctx[(size_t)flat * a.C + k], f);
    *reinterpret_cast<f32x4*>(a.h + (size_t)i * a.H + c) = acc;
    if (c == 0) {
        const f32x4 v = {row[0] * m, row[1] * m, row[2] * m, 0.0f};
        *reinterpret_cast<f32x4*>(a.x0 + (size_t)i * 4) = v;
        *reinterpret_cast<f32x4*>(a.xcur + (size_t)i * 4) = v;
        if (a.zero_max) { a.zero_max[2 * (size_t)i] = 0.f; a.zero_max[2 * (size_t)i + 1] = 0.f; }
    }
}

// ----------------------------------------------------------------------------- node GEMM (fp32 MFMA)
// C[M][Nc] = epi(A[M][K] * Wt[K][Nc] + bias).  Workgroup tile (32*WM) x (32*WN), one 32x32
// v_mfma_f32_32x32x2_f32 accumulator per wavefront, K in chunks of 32 double-buffered through LDS.
// The K index inside a chunk is permuted (lane half h owns k = 16h..16h+15) so that both
// operands are fetched with one ds_read_b128 per four MFMAs; weights are pre-packed in that image.

enum { EPI_BIAS = 0, EPI_BIAS_SILU = 1, EPI_RESID_MASK = 2,
       EPI_BIAS_MASK = 3,        // (acc + bias) * nmask[row]                                   (stage 2: new edge attributes * edge_mask)
       EPI_RANK1_SILU = 4,       // silu(rowv[row * rowv_stride] * colv[col] + (acc + bias))    (stage 2: the radial column of edge_mlp.0)
       EPI_EGCL_PRE = 5 };       // silu(A[erow] + B[ecol] + radial w_r + (acc + bias)), radial / unit direction written on the way
                                 // (stage 2, round 5: k_egcl_pre's expression in the epilogue of the edge-attribute GEMM)

struct GemmArgs {
    const float* A;       // [M_pad][lda], columns k < K1
    const float* A2;      // CAT: [M_pad][K - K1], columns k >= K1 (the aggregated neighbour messages)
    const float* Bimg;    // packed weight image
    const float* bias;    // [Nc]
    const float* nmask;   // [M_pad] (EPI_RESID_MASK, EPI_BIAS_MASK)
    float* C;             // [M_pad][ldc]
    int lda, ldc, K1, K, M, Nc;
    const float* rowv;    // EPI_RANK1_SILU: per-row scalar at rowv[row * rowv_stride] ...
    const float* colv;    // ... times the per-column vector colv[Nc]; EPI_EGCL_PRE: w_r
    int rowv_stride;
    const float* resid;   // EPI_RESID_MASK: residual rows [M][ldr] (NULL: the destination itself, in place)
    int ldr, resid_none;  // resid_none: no residual at all (stage 2, non-recurrent layer)
    // EPI_EGCL_PRE (rows = edges): node tables and where the geometry of an edge goes
    const int* erow; const int* ecol;     // [E] receiving / sending node of the edge
    const float* ABn;     // [M][2 Nc]: cols < Nc: W1a h + b1; cols >= Nc: W1b h
    const float* xn;      // [M][xs] coordinates
    float* geo;           // [E][4] = {cdiff_x, cdiff_y, cdiff_z, radial}, written by the column tile 0 workgroups
    int xs, geo_mode;
};


// WM x WN wavefronts, each owning 32 rows x (32*CN) columns (CN accumulators); workgroup tile
// (32*WM) x (32*WN*CN).  Used by the stage-2 layer (hd_egcl_forward); the sampler's node side is the fused k_node / k_node_f32.
// Weight image per (column tile, 32-wide K chunk): [NS][4 q][64 lanes][4 j], k = 32c + 16*(lane>>5) + 4q + j,
// with NS = WN*CN 32-column sub-tiles, column = tile*32*NS + 32*sub + (lane&31).
template <int WM, int WN, int CN, int EPI, bool CAT>
__global__ __launch_bounds__(WM * WN * 64) void k_gemm(GemmArgs g) {
    constexpr int NS = WN * CN;
    constexpr int BM = 32 * WM, BN = 32 * NS, NT = 64 * WM * WN;
    constexpr int A_F4 = BM * 8 / NT;                    // float4 of the A tile per thread
    constexpr int B_U4 = BN * 32 * 4 / 16 / NT;          // 16-byte pieces of the B image per thread
    constexpr int LDA_F = 36;                            // fp32 A row: 32 + 4 pad floats
    constexpr int A_BYTES = BM * LDA_F * 4;
    constexpr int B_BYTES = BN * 32 * 4;
    __shared__ __attribute__((aligned(16))) char smem_g[2 * (A_BYTES + B_BYTES)];
    auto As_f = [&](int buf) { return reinterpret_cast<float*>(smem_g + buf * (A_BYTES + B_BYTES)); };
    auto Bs = [&](int buf) { return smem_g + buf * (A_BYTES + B_BYTES) + A_BYTES; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int hh = lane >> 5, m = lane & 31;
    // XCD-aware tile order (1-D grid of 8 * ceil(nrt/8) * nct blocks, block b runs on XCD b % 8): every XCD owns
    // a contiguous range of row tiles and walks (row tile, column tile) with the column tile fastest, so the
    // A rows - written by the previous kernel, i.e. resident in Infinity Cache, not in this XCD's L2 - cross
    // the fabric once per XCD instead of once per column tile.  Speed only.
    int rt, ctile;
    {
        const int nrt = (g.M + BM - 1) / BM, nct = g.Nc / BN;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = nrt >> 3, r = nrt & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        if (idx >= len * nct) return;
        rt = start + idx / nct;
        ctile = idx % nct;
    }
    const int row0 = rt * BM;
    const int nchunk = g.K >> 5;
    const u32x4* Bsrc = reinterpret_cast<const u32x4*>(g.Bimg) + (size_t)ctile * nchunk * (B_BYTES / 16);

    // Global loads run three chunks ahead of the MFMAs (register ring), LDS is double-buffered: with only a
    // few workgroups per CU the ~1-2 us L2/MALL latency per chunk is otherwise exposed nchunk times.
    f32x4 ra3[3][A_F4];
    u32x4 rb3[3][B_U4];

    auto load_tiles = [&](int c, f32x4 (&ra)[A_F4], u32x4 (&rb)[B_U4]) {
        const int k0 = c << 5;
#pragma unroll
        for (int u = 0; u < A_F4; ++u) {
            int idx = tid + u * NT;
            int r = idx >> 3, sg = idx & 7;
            int row = row0 + r;
            if (row >= g.M) row = g.M - 1;          // rows past the end are never stored: read a valid one (A may be the caller's tensor)
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (!CAT || k0 < g.K1) v = *reinterpret_cast<const f32x4*>(g.A + (size_t)row * g.lda + k0 + 4 * sg);
            else v = *reinterpret_cast<const f32x4*>(g.A2 + (size_t)row * (g.K - g.K1) + (k0 - g.K1) + 4 * sg);
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < B_U4; ++u) rb[u] = Bsrc[(size_t)c * (B_BYTES / 16) + tid + u * NT];
    };
    auto store_tiles = [&](int buf, const f32x4 (&ra)[A_F4], const u32x4 (&rb)[B_U4]) {
#pragma unroll
        for (int u = 0; u < A_F4; ++u) {
            int idx = tid + u * NT;
            int r = idx >> 3, sg = idx & 7;
            *reinterpret_cast<f32x4*>(As_f(buf) + r * LDA_F + 4 * sg) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < B_U4; ++u) reinterpret_cast<u32x4*>(Bs(buf))[tid + u * NT] = rb[u];
    };

    f32x16 acc[CN];
#pragma unroll
    for (int cn = 0; cn < CN; ++cn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cn][r] = 0.f;

    auto compute = [&](int buf) {
        const float* Bf = reinterpret_cast<const float*>(Bs(buf));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(As_f(buf) + (32 * wr + m) * LDA_F + 16 * hh + 4 * q);
            f32x4 bv[CN];
#pragma unroll
            for (int cn = 0; cn < CN; ++cn)
                bv[cn] = *reinterpret_cast<const f32x4*>(Bf + (((wc * CN + cn) * 4 + q) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int cn = 0; cn < CN; ++cn)
                    acc[cn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[cn][j], acc[cn], 0, 0, 0);
        }
    };

    load_tiles(0, ra3[0], rb3[0]);
    if (nchunk > 1) load_tiles(1, ra3[1], rb3[1]);
    if (nchunk > 2) load_tiles(2, ra3[2], rb3[2]);
    store_tiles(0, ra3[0], rb3[0]);
    __syncthreads();
    // chunk c: compute from LDS[c&1]; stage chunk c+1 (ring slot (c+1)%3) into the other LDS buffer; refill
    // ring slot c%3 with chunk c+3.  Unrolled by 3 so the ring slots are compile-time.
    for (int c0 = 0; c0 < nchunk; c0 += 3) {
        static_for<0, 3>([&](auto Rc) {
            constexpr int rslot = decltype(Rc)::value;
            const int c = c0 + rslot;
            if (c < nchunk) {
                compute(c & 1);
                if (c + 1 < nchunk) store_tiles((c + 1) & 1, ra3[(rslot + 1) % 3], rb3[(rslot + 1) % 3]);
                if (c + 3 < nchunk) load_tiles(c + 3, ra3[rslot], rb3[rslot]);
                __syncthreads();
            }
        });
    }

    // Epilogue through LDS: the MFMA C layout gives each lane single floats of 16 different rows (16 dword
    // stores per accumulator, store-issue bound); transposed through the now idle staging buffers every
    // thread instead moves whole float4s (4x fewer, 16-byte wide, 256 B contiguous per 16 lanes).
    constexpr int LDC_S = BN + 4;
    static_assert(BM * LDC_S * 4 <= 2 * (A_BYTES + B_BYTES), "C tile must fit the staging buffers");
    float* Cs = reinterpret_cast<float*>(smem_g);
#pragma unroll
    for (int cn = 0; cn < CN; ++cn)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Cs[(32 * wr + (r & 3) + 8 * (r >> 2) + 4 * hh) * LDC_S + 32 * (wc * CN + cn) + m] = acc[cn][r];
    __syncthreads();
    constexpr int C_F4 = BM * BN / 4 / NT;
#pragma unroll
    for (int u = 0; u < C_F4; ++u) {
        const int idx = tid + u * NT;
        const int r = idx / (BN / 4), c4 = idx % (BN / 4);
        const int row = row0 + r, col = ctile * BN + 4 * c4;
        if (row < g.M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * LDC_S + 4 * c4) + *reinterpret_cast<const f32x4*>(g.bias + col);
            f32x4* dst = reinterpret_cast<f32x4*>(g.C + (size_t)row * g.ldc + col);
            if (EPI == EPI_BIAS_SILU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
            }
            if (EPI == EPI_RESID_MASK) {
                if (g.resid_none) v = v * g.nmask[row];                          // (0 + v) * mask: the same bits
                else if (g.resid) v = (*reinterpret_cast<const f32x4*>(g.resid + (size_t)row * g.ldr + col) + v) * g.nmask[row];
                else v = (*dst + v) * g.nmask[row];
            }
            if (EPI == EPI_EGCL_PRE) {              // k_egcl_pre's expressions, element by element, v = T1 row piece
                const int er = g.erow[row], ec = g.ecol[row];
                const float* xr = g.xn + (size_t)er * g.xs;
                const float* xc = g.xn + (size_t)ec * g.xs;
                const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dz = xr[2] - xc[2];
                const float radial = dx * dx + dy * dy + dz * dz;
                if (col == 0) {
                    const float inv = 1.0f / (sqrtf(radial + 1e-8f) + 1.0f);
                    *reinterpret_cast<f32x4*>(g.geo + (size_t)row * 4) = f32x4{dx * inv, dy * inv, dz * inv, radial};
                }
                f32x4 pre = *reinterpret_cast<const f32x4*>(g.ABn + (size_t)er * 2 * g.Nc + col) +
                            *reinterpret_cast<const f32x4*>(g.ABn + (size_t)ec * 2 * g.Nc + g.Nc + col);
                const f32x4 wr = *reinterpret_cast<const f32x4*>(g.colv + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(g.geo_mode ? 1.0f / (radial * radial) : radial, wr[j], pre[j]);
                pre += v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu_f(pre[j]);
            }
            if (EPI == EPI_BIAS_MASK) v = v * g.nmask[row];
            if (EPI == EPI_RANK1_SILU) {            // k_egcl_ew<0>'s expression, element by element
                const float rv = g.rowv[(size_t)row * g.rowv_stride];
                const f32x4 cw = *reinterpret_cast<const f32x4*>(g.colv + col);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu_f(__builtin_fmaf(rv, cw[j], v[j]));
            }
            *dst = v;
        }
    }
}

// ----------------------------------------------------------------------------- fused node update (bf16x3)
// One workgroup owns 32 node rows and runs the whole row-local chain of a GCL's node model plus the first
// edge Linear of the layer(s) that follow, so the intermediate activations never leave the CU:
//   X   = [h | (sum of the node's partial neighbour sums) / normalization_factor]      (egnn_new.py:52-56,280-282)
//   T   = silu(X W3^T + b3)                                                            (node_mlp.0 + SiLU, :58-66)
//   h'  = (h + T W4^T + b4) * mask                                                     (node_mlp.2, residual, mask)
//   AB_q = h' [W1a_q | W1b_q]^T + [b1_q | 0]   for the next NAB edge layers            (factorised edge_mlp.0 / coord_mlp.0)
// (UPD = false: only the last line, on h as it is - used once after the embedding.)  It replaces
// k_gemm(AB) + k_agg + k_gemm(n1) + k_gemm(n2): at M = 7,680 rows those four launches were bound by fixed costs
// (launch, tile prologue, C stores), not by math.
//   * A operands: the 32-row activation tile lives in LDS as bf16 head + tail, row stride K+8 elements
//     (16 B pad => conflict-free ds_read_b128), shared by all wavefronts.
//   * B operands: every wavefront owns its own 32-column tiles, so weights have no reuse inside a workgroup
//     and go L2 -> registers directly (fragment-ordered image, 1 KiB coalesced per load), PF k-steps ahead.
//   * C tiles leave through an LDS transpose as whole float4 rows.
// Weight image (pack_node_b): [k-step s][column tile ct][head|tail][64 lanes][8 bf16],
//   k = 16 s + 8 (lane>>5) + i,  col = 32 ct + (lane&31).
// NP = 3 is the same kernel in the bf16x6 arithmetic (k_edge.hpp): activations and weights in three bf16 pieces
// ([head|middle|tail] planes / image slots), six MFMAs per product, the fp32 mode's SiLU.

struct NodeArgs {
    const float* h_in;      // [M_pad][H]
    float* h_out;           // [M_pad][H] (may alias h_in: a workgroup only touches its own rows)
    const float* part;      // [P][H] partial neighbour sums of the edge kernel
    const int* pstart;      // [M+1]
    const float* nmask;     // [M_pad]
    const float* W3img;     // K = 2H, N = H
    const float* b3;
    const float* W4img;     // K = H, N = H
    const float* b4;
    const float* ABimg[2];  // K = H, N = 2H
    const float* ABbias[2]; // [2H]
    float* ABout[2];        // [M_pad][2H]
    float* ABmax[2];        // optional [M_pad][2]: max_k |A_i[k]|, max_k |B_i[k]| of the AB rows written (fp16x3 edge kernels)
    // F16 variant (two-piece FP16 operands, k_node<..., 2, true>): 1 / (power-of-two scale) of the three weight images, and the
    // constants of the a-priori row bounds: max_c sum_k |W[c][k]| of W3 / W4 and max |b3| / |b4|
    float w3inv, w4inv, abinv[2], w3l1, w4l1, b3max, b4max;
    float norm;
    int M;
};

// acc[c] += A[32 x 16 KS] * B[:, column tile ct(c)]   with ct(c) = (c / CTW) * CTG + ct0 + c % CTW.
// B fragments travel L2 -> registers in a ring of PF k-steps; `prefetch` fills the ring (it is issued before
// the barrier / epilogue that precedes the contraction, weights do not depend on data) and `run` consumes
// it.  sched_barrier(0) at every k-step keeps hipcc from sinking the loads next to their MFMAs (it otherwise
// shrinks the ring to 2-3 loads in flight to save registers and exposes the L2 latency every k-step).
template <int KS, int CTn, int CTW, int PF, int NCT, int NP = 2, bool F16 = false>
struct NodeMma {
    typedef u32x4 Ring[PF][CTn][NP];
    template <int s, int slot>
    static HD_DEVINL void load(Ring& br, const u32x4* Bl, int ct0, int CTG) {
#pragma unroll
        for (int c = 0; c < CTn; ++c) {
            const int ct = (c / CTW) * CTG + ct0 + c % CTW;
#pragma unroll
            for (int p = 0; p < NP; ++p) br[slot][c][p] = Bl[((size_t)(s * NCT + ct) * NP + p) * 64];
        }
    }
    static HD_DEVINL void prefetch(Ring& br, const u32x4* Bl, int ct0, int CTG) {
        static_for<0, (PF < KS ? PF : KS)>([&](auto S) { load<decltype(S)::value, decltype(S)::value>(br, Bl, ct0, CTG); });
        asm volatile("" ::: "memory");                // keeps the loads above whatever follows (barriers included)
        __builtin_amdgcn_sched_barrier(0);
    }
    // Ap[p]: this lane's row of piece p of the A tile (head, [middle,] tail).  Two pieces: a_h b_h + a_l b_h + a_h b_l (bf16x3);
    // three pieces: a_h b_l + a_l b_h + a_m b_m + a_h b_m + a_m b_h + a_h b_h, small terms first (bf16x6, k_edge.hpp).
    // F16: the K range is summed in four QUARTERS with their own accumulators, result = ((q0 + q1) + q2) + q3 - the order of
    // k_node_split.hpp, whose four wavefronts per output tile own one quarter each (the two paths are bit-identical).  `acc`
    // arrives zeroed in that mode (the bias joins in the un-scaling fma) and serves as q0.
    static constexpr int NQ = F16 ? 4 : 1;
    static HD_DEVINL void run(f32x16 (&acc)[CTn], Ring& br, const __bf16* const (&Ap)[NP], const u32x4* Bl, int ct0, int CTG) {
        static_assert(KS % NQ == 0, "whole k-steps per quarter");
        f32x16 accq[NQ > 1 ? NQ - 1 : 1][CTn];
        if constexpr (NQ > 1) {
#pragma unroll
            for (int q = 0; q < NQ - 1; ++q)
#pragma unroll
                for (int c = 0; c < CTn; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accq[q][c][r] = 0.f;
        }
        bf16x8_t a[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = *reinterpret_cast<const bf16x8_t*>(Ap[p]);
        static_for<0, KS>([&](auto S) {
            constexpr int s = decltype(S)::value, slot = s % PF;
            constexpr int qi = s / (KS / NQ);
            f32x16(&dst)[CTn] = *(qi == 0 ? &acc : &accq[qi > 0 ? qi - 1 : 0]);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t an[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) an[p] = a[p];
            if constexpr (s + 1 < KS) {
#pragma unroll
                for (int p = 0; p < NP; ++p) an[p] = *reinterpret_cast<const bf16x8_t*>(Ap[p] + 16 * (s + 1));
            }
            __builtin_amdgcn_sched_barrier(0);         // next A fragments are in flight under this step's MFMAs
            bf16x8_t b[CTn][NP];
#pragma unroll
            for (int c = 0; c < CTn; ++c)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[c][p] = __builtin_bit_cast(bf16x8_t, br[slot][c][p]);
            auto term = [&](int pa, int pb) {
#pragma unroll
                for (int c = 0; c < CTn; ++c) {
                    if constexpr (F16) dst[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[pa]), __builtin_bit_cast(f16x8, b[c][pb]), dst[c], 0, 0, 0);
                    else dst[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[c][pb], dst[c], 0, 0, 0);
                }
            };
            if constexpr (NP == 2) { term(0, 0); term(1, 0); term(0, 1); }
            else { term(0, 2); term(2, 0); term(1, 1); term(0, 1); term(1, 0); term(0, 0); }
            if constexpr (s + PF < KS) load<s + PF, slot>(br, Bl, ct0, CTG);
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = an[p];
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NQ > 1) {
#pragma unroll
            for (int c = 0; c < CTn; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = ((acc[c][r] + accq[0][c][r]) + accq[1][c][r]) + accq[2][c][r];
        }
    }
};

// v -> NP bf16 pieces at the same element offset of NP consecutive LDS planes of `plane` elements
template <int NP>
HD_DEVINL void bf16_split_store(__bf16* d, int plane, float v) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const __bf16 piece = (__bf16)v;
        d[p * plane] = piece;
        v -= (float)piece;
    }
}

// the same with FP16 pieces (two planes; the value arrives scaled into range, see k_node<..., F16>)
HD_DEVINL void f16_split_store(__bf16* d, int plane, float v) {
    _Float16* p = reinterpret_cast<_Float16*>(d);
    const _Float16 hi = (_Float16)v;
    p[0] = hi;
    p[plane] = (_Float16)(v - (float)hi);
}
// power-of-two scale s with bound * s in [2^13, 2^14) and its inverse (k_edge.hpp, fp16x3)
HD_DEVINL void f16_row_scale(float bound, float& s, float& inv) {
    const uint32_t eb = (__builtin_bit_cast(uint32_t, bound + HD_F16_FLOOR) >> 23) & 0xffu;
    inv = __builtin_bit_cast(float, (eb - 13u) << 23);
    s = __builtin_bit_cast(float, (267u - eb) << 23);
}

// F16 (NPC = 2 only): the fp16x3 arithmetic of the edge kernels for the three node GEMMs - two-way FP16 split, three fp16 MFMAs per
// product, operands ranged by exact powers of two.  Weights: per matrix, by the packer.  Activations: per ROW, from bounds that are
// known before the first contraction starts (so no reduction across wavefronts is needed between the phases):
//     X = [h | agg]:  max_k |X_r[k]|                                   (16 threads hold a row in phase 0: four shuffles)
//     T = SiLU(X W3^T + b3):  |T_r| <= max|X_r| max_c sum_k|W3[c][k]| + max|b3|
//     h' = (h + T W4^T + b4) mask:  |h'_r| <= max|h_r| + bound(T_r) max_c sum_k|W4[c][k]| + max|b4|
// The bounds are loose (an L1 norm against a random-sign sum: ~16 x for T, ~128 x for h'), which costs nothing: a scaled operand
// keeps 22 significant bits down to 2^-2, i.e. over 15 binades below its bound, and smaller elements keep an absolute error of
// 2^-25 / scale.  Each phase's accumulators hold (row scale x weight scale) x the result; the epilogue undoes it in the fma that
// adds the bias.
template <int H, int NW, bool UPD, int NAB, int NPC = 2, bool F16 = false>
__global__ __launch_bounds__(64 * NW, 1) void k_node(NodeArgs a) {
    static_assert(!F16 || NPC == 2, "the FP16 variant is a two-piece split");
    constexpr int NT = 64 * NW;
    constexpr int NCT = H / 32;            // column tiles of an H-wide output
    constexpr int CT = NCT / NW;           // ... per wavefront
    static_assert(CT >= 1 && CT * NW == NCT, "NW must divide H/32");
    constexpr int KX = UPD ? 2 * H : H;
    constexpr int LDX = KX + 8, LDH = H + 8;
    constexpr int PF12 = 4, PF3 = 3;       // k-steps of weights in flight per wavefront (deeper rings measured no faster)
    constexpr int R0_BYTES = 32 * LDX * 2 * NPC;        // the NP bf16 pieces of X (NPC = 2: head + tail; 3: bf16x6 mode)
    constexpr int PX = 32 * LDX, PH = 32 * LDH;        // elements per piece plane
    extern __shared__ __attribute__((aligned(16))) char smem_n[];
    __bf16* Xh = reinterpret_cast<__bf16*>(smem_n);                 // planes Xh + p * PX
    __bf16* Th = reinterpret_cast<__bf16*>(smem_n + R0_BYTES);      // region 1: T (planes + p * PH), later the AB staging tile
    __bf16* Nh = reinterpret_cast<__bf16*>(smem_n);                 // h' pieces re-use region 0 (planes + p * PH) ...
    float* stage0 = reinterpret_cast<float*>(smem_n + PH * 2 * NPC); // ... followed by its fp32 staging tile [32][H]
    constexpr int LDS1 = H + 4;
    float* stage1 = reinterpret_cast<float*>(smem_n + R0_BYTES);    // [32][H+4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    // XCD-aware row-tile order: block b runs on XCD b % 8; every XCD owns a contiguous range of row tiles, the
    // same split the edge kernel uses for its edge list, so `part` / `AB` rows stay in the XCD that touches them.
    int rt;
    {
        const int nrt = (a.M + 31) >> 5;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = nrt >> 3, r = nrt & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        if (idx >= len) return;
        rt = start + idx;
    }
    const int row0 = rt * 32;

    typedef NodeMma<KX / 16, CT, CT, PF12, NCT, NPC, F16> M1;        // X W3^T      (UPD only)
    typedef NodeMma<H / 16, CT, CT, PF12, NCT, NPC, F16> M2;         // T W4^T      (UPD only)
    typedef NodeMma<H / 16, 2 * CT, CT, PF3, 2 * NCT, NPC, F16> M3;  // h' [W1a|W1b]^T
    // F16: per row of the tile {scale of T, scale of h', 1 / scale of X, 1 / scale of T, 1 / scale of h'}
    __shared__ __attribute__((aligned(16))) float rsc[5][32];
    auto row4 = [&](int k, int q) { return *reinterpret_cast<const f32x4*>(&rsc[k][8 * q + 4 * hh]); };
    auto rows_of = [&](const __bf16* base, int ld, int plane, const __bf16* (&out)[NPC]) {
#pragma unroll
        for (int p = 0; p < NPC; ++p) out[p] = base + p * plane + n * ld + 8 * hh;
    };
    typename M1::Ring br1;
    typename M2::Ring br2;
    typename M3::Ring br3;
    const int ct0 = wave * CT;
    const u32x4* W3l = reinterpret_cast<const u32x4*>(a.W3img) + lane;
    const u32x4* W4l = reinterpret_cast<const u32x4*>(a.W4img) + lane;
    const u32x4* AB0l = reinterpret_cast<const u32x4*>(a.ABimg[0]) + lane;
    if constexpr (UPD) M1::prefetch(br1, W3l, ct0, 0);
    // biases of every phase, requested at kernel entry (a load at its point of use is an exposed L2 round trip per phase)
    float b3v[CT], b4v[CT], abv[NAB][2 * CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        b3v[c] = UPD ? a.b3[32 * (ct0 + c) + n] : 0.f;
        b4v[c] = UPD ? a.b4[32 * (ct0 + c) + n] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NAB; ++q)
#pragma unroll
        for (int c = 0; c < 2 * CT; ++c) abv[q][c] = a.ABbias[q][(c / CT) * H + 32 * (ct0 + c % CT) + n];

    // ---- phase 0: X -> LDS (bf16 head/tail).  NT/32 threads per row, each moving every (NT/32)-th float4 of
    // the row, so a thread needs one pstart pair and all its loads are independent of each other.
    {
        constexpr int Q = H / 4;                     // float4 per H-wide row
        constexpr int TPR = NT / 32;                 // threads per row
        constexpr int NP = Q / TPR;                  // pieces per thread and source
        static_assert(Q % TPR == 0, "row pieces must divide evenly");
        const int r = tid / TPR, cq = tid % TPR;
        const int row = row0 + r;
        float sX = 1.0f;                                  // F16: this row's scale of X
        auto put = [&](int col, f32x4 v) {
            if constexpr (F16) {
                typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
                v = f32x4{v[0] * sX, v[1] * sX, v[2] * sX, v[3] * sX};
                const f16x4_t hi = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                const f16x4_t lo = {(_Float16)(v[0] - (float)hi[0]), (_Float16)(v[1] - (float)hi[1]), (_Float16)(v[2] - (float)hi[2]),
                                    (_Float16)(v[3] - (float)hi[3])};
                *reinterpret_cast<f16x4_t*>(Xh + r * LDX + col) = hi;
                *reinterpret_cast<f16x4_t*>(Xh + PX + r * LDX + col) = lo;
                return;
            }
#pragma unroll
            for (int p = 0; p < NPC; ++p) {
                const __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
                *reinterpret_cast<bf16x4_t*>(Xh + p * PX + r * LDX + col) = bf16x4_t{h0, h1, h2, h3};
                v = f32x4{v[0] - (float)h0, v[1] - (float)h1, v[2] - (float)h2, v[3] - (float)h3};
            }
        };
        int p0 = 0, p1 = 0;
        if constexpr (UPD) {
            if (row < a.M) { p0 = a.pstart[row]; p1 = a.pstart[row + 1]; }
        }
        f32x4 hv[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u)
            hv[u] = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)row * H + 4 * (cq + u * TPR));   // pad rows are zero
        if constexpr (UPD) {
            // the first two partial sums (the common case: a node's edges span two tiles) are fetched together
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
            if constexpr (F16) {
                f32x4 gv[NP];
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
                const float tb = __builtin_fmaf(mx, a.w3l1, a.b3max), hb = mh + __builtin_fmaf(tb, a.w4l1, a.b4max);
                float iX, sT, iT, sH, iH;
                f16_row_scale(mx, sX, iX); f16_row_scale(tb, sT, iT); f16_row_scale(hb, sH, iH);
                if (cq == 0) { rsc[0][r] = sT; rsc[1][r] = sH; rsc[2][r] = iX * a.w3inv; rsc[3][r] = iT * a.w4inv; rsc[4][r] = iH; }
#pragma unroll
                for (int u = 0; u < NP; ++u) put(4 * (cq + u * TPR), hv[u]);
#pragma unroll
                for (int u = 0; u < NP; ++u) put(H + 4 * (cq + u * TPR), gv[u]);
            } else {
#pragma unroll
            for (int u = 0; u < NP; ++u) put(4 * (cq + u * TPR), hv[u]);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                f32x4 v = z4;
                if (has0) v += g0[u];
                if (has1) v += g1[u];
                for (int p = p0 + 2; p < p1; ++p) v += *reinterpret_cast<const f32x4*>(a.part + (size_t)p * H + 4 * (cq + u * TPR));
                put(H + 4 * (cq + u * TPR), v / a.norm);
            }
            }
        } else {
            if constexpr (F16) {                                     // AB only: X = h, the operand of phase 3
                float mh = 0.f;
#pragma unroll
                for (int u = 0; u < NP; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mh = fmaxf(mh, fabsf(hv[u][j]));
#pragma unroll
                for (int o = TPR / 2; o > 0; o >>= 1) mh = fmaxf(mh, __shfl_xor(mh, o));
                float iX;
                f16_row_scale(mh, sX, iX);
                if (cq == 0) rsc[4][r] = iX;
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) put(4 * (cq + u * TPR), hv[u]);
        }
    }

    if constexpr (UPD) {
        // ---- phase 1: T = silu(X W3^T + b3)
        {
            f32x16 acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float b = F16 ? 0.f : b3v[c];                  // F16: the bias joins in the un-scaling fma
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = b;
            }
            __syncthreads();                                         // X complete
            M2::prefetch(br2, W4l, ct0, 0);
            const __bf16* xr[NPC];
            rows_of(Xh, LDX, PX, xr);
            M1::run(acc, br1, xr, W3l, ct0, 0);
            if constexpr (F16) {
                f32x4 un[4], sc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { un[q] = row4(2, q); sc[q] = row4(0, q); }
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int R = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const float t = silu_f(__builtin_fmaf(acc[c][r], un[r >> 2][r & 3], b3v[c]));
                        f16_split_store(Th + R * LDH + 32 * (ct0 + c) + n, PH, t * sc[r >> 2][r & 3]);
                    }
            } else {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    // bf16x3: plain SiLU (the contraction error is ~1e-6 anyway); bf16x6: the fp32 mode's compensated one
                    bf16_split_store<NPC>(Th + R * LDH + 32 * (ct0 + c) + n, PH, NPC == 3 ? silu_f(acc[c][r]) : silu_fast(acc[c][r]));
                }
            }
        }
        __syncthreads();
        // ---- phase 2: h' = (h + T W4^T + b4) * mask
        {
            f32x16 acc[CT];
            float hres[CT][16], mk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mk[r] = a.nmask[row0 + (r & 3) + 8 * (r >> 2) + 4 * hh];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float b = F16 ? 0.f : b4v[c];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[c][r] = b;
                    hres[c][r] = a.h_in[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * H + 32 * (ct0 + c) + n];
                }
            }
            M3::prefetch(br3, AB0l, ct0, NCT);
            const __bf16* tr[NPC];
            rows_of(Th, LDH, PH, tr);
            M2::run(acc, br2, tr, W4l, ct0, 0);
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if constexpr (F16) {
                        const float v = (hres[c][r] + __builtin_fmaf(acc[c][r], row4(3, r >> 2)[r & 3], b4v[c])) * mk[r];
                        f16_split_store(Nh + R * LDH + 32 * (ct0 + c) + n, PH, v * row4(1, r >> 2)[r & 3]);
                        stage0[R * H + 32 * (ct0 + c) + n] = v;
                    } else {
                    const float v = (hres[c][r] + acc[c][r]) * mk[r];
                    bf16_split_store<NPC>(Nh + R * LDH + 32 * (ct0 + c) + n, PH, v);
                    stage0[R * H + 32 * (ct0 + c) + n] = v;
                    }
                }
        }
        __syncthreads();
        {
            constexpr int Q = H / 4, NP = 32 * Q / NT;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int idx = tid + u * NT, r = idx / Q, c4 = idx % Q;
                if (row0 + r < a.M)
                    *reinterpret_cast<f32x4*>(a.h_out + (size_t)(row0 + r) * H + 4 * c4) = *reinterpret_cast<const f32x4*>(stage0 + r * H + 4 * c4);
            }
        }
    }

    // ---- phase 3: AB_q = h' [W1a | W1b]^T + bias, two H-wide halves per wavefront, staged through region 1
#pragma unroll
    for (int q = 0; q < NAB; ++q) {
        f32x16 acc[2 * CT];
#pragma unroll
        for (int c = 0; c < 2 * CT; ++c) {
            const float b = F16 ? 0.f : abv[q][c];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = b;
        }
        const u32x4* ABl = reinterpret_cast<const u32x4*>(a.ABimg[q]) + lane;
        if (q > 0 || !UPD) {
            M3::prefetch(br3, ABl, ct0, NCT);
            if (!UPD) __syncthreads();                               // h tile complete
        }
        const __bf16* nr[NPC];
        rows_of(Nh, LDH, PH, nr);                            // (!UPD: h' is X itself, LDX == LDH)
        M3::run(acc, br3, nr, ABl, ct0, NCT);
        if constexpr (F16) {
            f32x4 un[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) un[qq] = row4(4, qq);
            const float wi = a.abinv[q];
#pragma unroll
            for (int c = 0; c < 2 * CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = __builtin_fmaf(acc[c][r], un[r >> 2][r & 3] * wi, abv[q][c]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half || q) __syncthreads();                 // previous staging tile fully stored
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage1[((r & 3) + 8 * (r >> 2) + 4 * hh) * LDS1 + 32 * (ct0 + c) + n] = acc[half * CT + c][r];
            __syncthreads();
            constexpr int Q = H / 4, NP = 32 * Q / NT;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int idx = tid + u * NT, r = idx / Q, c4 = idx % Q;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage1 + r * LDS1 + 4 * c4);
                if (row0 + r < a.M) *reinterpret_cast<f32x4*>(a.ABout[q] + (size_t)(row0 + r) * 2 * H + half * H + 4 * c4) = v;
                if constexpr (Q == 64 || Q == 32) {
                    // Q consecutive lanes store one complete row of the half per pass (a wavefront at width 256, half of one at
                    // 128): its maximum |value| is a lane reduction away (the per-node part of the fp16x3 edge kernels'
                    // activation bound, k_edge.hpp)
                    if (a.ABmax[q]) {
                        float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
                        for (int o = Q / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                        if ((lane & (Q - 1)) == 0 && row0 + r < a.M) a.ABmax[q][2 * (size_t)(row0 + r) + half] = m;
                    }
                }
            }
        }
    }
}

// Measurement build only: per-wave cycle stamps of the LAST k_node_f32 launch (hd_debug_node_trace, scratch/node_trace.py).
#ifdef HD_DEBUG_KERNELS
#define HD_NTRACE_STAMPS 12
__device__ long long hd_ntrace[512 * 8 * HD_NTRACE_STAMPS];
#define HD_NSTAMP(k)                                                                                                   \
    do {                                                                                                               \
        const long long ts_ = __builtin_readcyclecounter();                                                            \
        if (lane == 0 && blockIdx.x < 512) hd_ntrace[((size_t)blockIdx.x * 8 + wave) * HD_NTRACE_STAMPS + (k)] = ts_;    \
    } while (0)
#else
#define HD_NSTAMP(k) do { } while (0)
#endif

// ----------------------------------------------------------------------------- fused node update, exact fp32
// The same launch structure as k_node (one launch per node update: neighbour-sum reduction, node MLP, residual, the next
// layers' first edge Linear) on v_mfma_f32_32x32x2_f32 - the fp32 mode's node side (round 1 and most of round 2 ran it as
// k_agg + 3 x k_gemm: 62-83 us per update at M = 7,680 rows, bound by launch / tile-prologue costs rather than math).
//   * the 32-row activation tile lives in LDS as fp32, row stride K+4 floats (conflict-free ds_read_b128);
//   * weights go L2 -> registers in fragment order, per 32-wide K chunk and column tile [4 q][64 lanes][4 j] floats with
//     k = 32 s + 16 (lane>>5) + 4 q + j, col = 32 ct + (lane&31)   (pack_node_b_f32), a ring of PF chunks per wavefront.
// NQ = 4 (widths >= 128, round 5): the K range is summed in four QUARTERS with their own accumulators, result = ((q0 + q1) + q2) + q3 -
// the order of k_node_split_f32 (k_node_split.hpp), whose four wavefronts per output tile own one quarter each.  `acc` arrives
// zeroed and serves as q0.  NQ = 1: one chain (narrow widths, whose small-batch twin is k_gemm_r16).
template <int KS, int CTn, int CTW, int PF, int NCT, int NQ = 1>
struct NodeMmaF {
    typedef u32x4 Ring[PF][CTn][4];
    template <int s, int slot>
    static HD_DEVINL void load(Ring& br, const u32x4* Bl, int ct0, int CTG) {
#pragma unroll
        for (int c = 0; c < CTn; ++c) {
            const int ct = (c / CTW) * CTG + ct0 + c % CTW;
#pragma unroll
            for (int q = 0; q < 4; ++q) br[slot][c][q] = Bl[((size_t)(s * NCT + ct) * 4 + q) * 64];
        }
    }
    static HD_DEVINL void prefetch(Ring& br, const u32x4* Bl, int ct0, int CTG) {
        static_for<0, (PF < KS ? PF : KS)>([&](auto S) { load<decltype(S)::value, decltype(S)::value>(br, Bl, ct0, CTG); });
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    static HD_DEVINL void run(f32x16 (&acc)[CTn], Ring& br, const float* Arow, const u32x4* Bl, int ct0, int CTG) {
        static_assert(KS % NQ == 0, "whole K chunks per quarter");
        f32x16 accq[NQ > 1 ? NQ - 1 : 1][CTn];
        if constexpr (NQ > 1) {
#pragma unroll
            for (int q = 0; q < NQ - 1; ++q)
#pragma unroll
                for (int c = 0; c < CTn; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accq[q][c][r] = 0.f;
        }
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const f32x4*>(Arow + 4 * q);
        static_for<0, KS>([&](auto S) {
            constexpr int s = decltype(S)::value, slot = s % PF;
            constexpr int qi = s / (KS / NQ);
            f32x16(&dst)[CTn] = *(qi == 0 ? &acc : &accq[qi > 0 ? qi - 1 : 0]);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 an[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) an[q] = a[q];
            if constexpr (s + 1 < KS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) an[q] = *reinterpret_cast<const f32x4*>(Arow + 32 * (s + 1) + 4 * q);
            }
            __builtin_amdgcn_sched_barrier(0);         // next A fragments are in flight under this chunk's MFMAs
            f32x4 b[CTn][4];
#pragma unroll
            for (int c = 0; c < CTn; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) b[c][q] = __builtin_bit_cast(f32x4, br[slot][c][q]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < CTn; ++c) dst[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], b[c][q][j], dst[c], 0, 0, 0);
            if constexpr (s + PF < KS) load<s + PF, slot>(br, Bl, ct0, CTG);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = an[q];
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NQ > 1) {
#pragma unroll
            for (int c = 0; c < CTn; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = ((acc[c][r] + accq[0][c][r]) + accq[1][c][r]) + accq[2][c][r];
        }
    }
};

template <int H, int NW, bool UPD, int NAB>
__global__ __launch_bounds__(64 * NW, 1) void k_node_f32(NodeArgs a) {
    constexpr int NT = 64 * NW;
    constexpr int NCT = H / 32;            // column tiles of an H-wide output
    constexpr int CT = NCT / NW;           // ... per wavefront
    static_assert(CT >= 1 && CT * NW == NCT, "NW must divide H/32");
    constexpr int KX = UPD ? 2 * H : H;
    constexpr int LDX = KX + 4, LDH = H + 4;
    constexpr int PF12 = 2, PF3 = 2;       // 32-wide K chunks of weights in flight per wavefront (3: no faster, 4: spills)
    constexpr int R0_BYTES = 32 * LDX * 4;
    extern __shared__ __attribute__((aligned(16))) char smem_n[];
    float* X = reinterpret_cast<float*>(smem_n);                    // region 0: X = [h | agg], later h'
    float* T = reinterpret_cast<float*>(smem_n + R0_BYTES);         // region 1: T, later the AB staging tile (same stride)
    float* Nn = X;                                                  // h' [32][LDH] (UPD) - or X itself (stride LDX == LDH)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    int rt;                                                         // XCD-aware row-tile order, as in k_node
    {
        const int nrt = (a.M + 31) >> 5;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = nrt >> 3, r = nrt & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        if (idx >= len) return;
        rt = start + idx;
    }
    const int row0 = rt * 32;
    HD_NSTAMP(0);

    // widths >= 128 sum every contraction in four K quarters (NodeMmaF, bit-identical to k_node_split_f32); their AB phase then
    // runs one H-wide half at a time (four quarter accumulators per column tile: two halves at once would not fit the registers)
    constexpr int NQ = H >= 128 ? 4 : 1;
    constexpr bool HALVES = NQ > 1;
    typedef NodeMmaF<KX / 32, CT, CT, PF12, NCT, NQ> M1;            // X W3^T      (UPD only)
    typedef NodeMmaF<H / 32, CT, CT, PF12, NCT, NQ> M2;             // T W4^T      (UPD only)
    typedef NodeMmaF<H / 32, (HALVES ? CT : 2 * CT), CT, (HALVES ? 2 * PF3 : PF3), 2 * NCT, NQ> M3;      // h' [W1a|W1b]^T (HALVES: one half per run)
    typename M1::Ring br1;
    typename M2::Ring br2;
    typename M3::Ring br3;
    const int ct0 = wave * CT;
    const u32x4* W3l = reinterpret_cast<const u32x4*>(a.W3img) + lane;
    const u32x4* W4l = reinterpret_cast<const u32x4*>(a.W4img) + lane;
    const u32x4* AB0l = reinterpret_cast<const u32x4*>(a.ABimg[0]) + lane;
    if constexpr (UPD) M1::prefetch(br1, W3l, ct0, 0);
    // biases of every phase, requested at kernel entry (a load at its point of use is an exposed L2 round trip per phase)
    float b3v[CT], b4v[CT], abv[NAB][2 * CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        b3v[c] = UPD ? a.b3[32 * (ct0 + c) + n] : 0.f;
        b4v[c] = UPD ? a.b4[32 * (ct0 + c) + n] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NAB; ++q)
#pragma unroll
        for (int c = 0; c < 2 * CT; ++c) abv[q][c] = a.ABbias[q][(c / CT) * H + 32 * (ct0 + c % CT) + n];

    // ---- phase 0: X -> LDS.  NT/32 threads per row, each moving every (NT/32)-th float4 of the row.
    {
        constexpr int Q = H / 4;                     // float4 per H-wide row
        constexpr int TPR = NT / 32;                 // threads per row
        constexpr int NPT = Q / TPR;                 // pieces per thread and source
        static_assert(Q % TPR == 0, "row pieces must divide evenly");
        const int r = tid / TPR, cq = tid % TPR;
        const int row = row0 + r;
        int p0 = 0, p1 = 0;
        if constexpr (UPD) {
            if (row < a.M) { p0 = a.pstart[row]; p1 = a.pstart[row + 1]; }
        }
        f32x4 hv[NPT];
#pragma unroll
        for (int u = 0; u < NPT; ++u)
            hv[u] = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)row * H + 4 * (cq + u * TPR));   // pad rows are zero
        if constexpr (UPD) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const bool has0 = p0 < p1, has1 = p0 + 1 < p1;
            const float* s0 = a.part + (size_t)(has0 ? p0 : 0) * H;
            const float* s1 = a.part + (size_t)(has1 ? p0 + 1 : 0) * H;
            f32x4 g0[NPT], g1[NPT];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                g0[u] = *reinterpret_cast<const f32x4*>(s0 + 4 * (cq + u * TPR));
                g1[u] = *reinterpret_cast<const f32x4*>(s1 + 4 * (cq + u * TPR));
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) *reinterpret_cast<f32x4*>(X + r * LDX + 4 * (cq + u * TPR)) = hv[u];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                f32x4 v = z4;                        // same order of additions as k_agg: parts ascending, then / norm
                if (has0) v += g0[u];
                if (has1) v += g1[u];
                for (int p = p0 + 2; p < p1; ++p) v += *reinterpret_cast<const f32x4*>(a.part + (size_t)p * H + 4 * (cq + u * TPR));
                *reinterpret_cast<f32x4*>(X + r * LDX + H + 4 * (cq + u * TPR)) = v / a.norm;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NPT; ++u) *reinterpret_cast<f32x4*>(X + r * LDX + 4 * (cq + u * TPR)) = hv[u];
        }
    }

    HD_NSTAMP(1);
    if constexpr (UPD) {
        // ---- phase 1: T = silu(X W3^T + b3)
        {
            // accumulators start at zero and the bias is added after the contraction, like k_gemm's epilogue: together with the
            // same MFMA order per output element this makes the fused kernel bit-identical to the k_agg + k_gemm chain
            f32x16 acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
            __syncthreads();                                         // X complete
            HD_NSTAMP(2);
            M2::prefetch(br2, W4l, ct0, 0);
            M1::run(acc, br1, X + n * LDX + 16 * hh, W3l, ct0, 0);
            HD_NSTAMP(3);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float b = b3v[c];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[((r & 3) + 8 * (r >> 2) + 4 * hh) * LDH + 32 * (ct0 + c) + n] = silu_f(acc[c][r] + b);
            }
        }
        __syncthreads();
        HD_NSTAMP(4);
        // ---- phase 2: h' = (h + T W4^T + b4) * mask
        {
            f32x16 acc[CT];
            float hres[CT][16], mk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) mk[r] = a.nmask[row0 + (r & 3) + 8 * (r >> 2) + 4 * hh];
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[c][r] = 0.f;
                    hres[c][r] = a.h_in[(size_t)(row0 + (r & 3) + 8 * (r >> 2) + 4 * hh) * H + 32 * (ct0 + c) + n];
                }
            M3::prefetch(br3, AB0l, ct0, NCT);
            M2::run(acc, br2, T + n * LDH + 16 * hh, W4l, ct0, 0);
            HD_NSTAMP(5);
            // every wave is done reading X once it is past its own M1::run AND the barrier above: h' may overwrite region 0
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float b = b4v[c];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    Nn[R * LDH + 32 * (ct0 + c) + n] = (hres[c][r] + (acc[c][r] + b)) * mk[r];
                }
            }
        }
        __syncthreads();
        {
            constexpr int Q = H / 4, NPT = 32 * Q / NT;
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int idx = tid + u * NT, r = idx / Q, c4 = idx % Q;
                if (row0 + r < a.M)
                    *reinterpret_cast<f32x4*>(a.h_out + (size_t)(row0 + r) * H + 4 * c4) = *reinterpret_cast<const f32x4*>(Nn + r * LDH + 4 * c4);
            }
        }
    }

    HD_NSTAMP(6);
    // ---- phase 3: AB_q = h' [W1a | W1b]^T + bias, two H-wide halves per wavefront, staged through region 1
#pragma unroll
    for (int q = 0; q < NAB; ++q) {
        f32x16 acc[2 * CT];
#pragma unroll
        for (int c = 0; c < 2 * CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        const u32x4* ABl = reinterpret_cast<const u32x4*>(a.ABimg[q]) + lane;
        if ((q > 0 && !HALVES) || !UPD) {                            // (HALVES: image q's first chunks were requested behind image q-1's last half)
            M3::prefetch(br3, ABl, ct0, NCT);
            if (!UPD) __syncthreads();                               // h tile complete
        }
        if constexpr (!HALVES) M3::run(acc, br3, Nn + n * LDH + 16 * hh, ABl, ct0, NCT);
        HD_NSTAMP(7 + 2 * q);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if constexpr (HALVES) {
                // this half's contraction (its first chunks were requested before the previous half was staged)
                f32x16(&ah)[CT] = *reinterpret_cast<f32x16(*)[CT]>(&acc[half * CT]);
                M3::run(ah, br3, Nn + n * LDH + 16 * hh, ABl, ct0 + half * NCT, 0);
                if (half == 0) M3::prefetch(br3, ABl, ct0 + NCT, 0);
                else if (q + 1 < NAB) M3::prefetch(br3, reinterpret_cast<const u32x4*>(a.ABimg[q + 1 < NAB ? q + 1 : q]) + lane, ct0, 0);
            }
            if (half || q) __syncthreads();                 // previous staging tile fully stored
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float b = abv[q][half * CT + c];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[((r & 3) + 8 * (r >> 2) + 4 * hh) * LDH + 32 * (ct0 + c) + n] = acc[half * CT + c][r] + b;
            }
            __syncthreads();
            constexpr int Q = H / 4, NPT = 32 * Q / NT;
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const int idx = tid + u * NT, r = idx / Q, c4 = idx % Q;
                if (row0 + r < a.M)
                    *reinterpret_cast<f32x4*>(a.ABout[q] + (size_t)(row0 + r) * 2 * H + half * H + 4 * c4) =
                        *reinterpret_cast<const f32x4*>(T + r * LDH + 4 * c4);
            }
        }
        HD_NSTAMP(8 + 2 * q);
    }
    HD_NSTAMP(11);
}
