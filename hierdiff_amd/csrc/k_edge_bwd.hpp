// Training kernels of the edge model (exact fp32): backward of one edge layer - gather, first edge Linear (factorised),
// SiLU, H x H Linear on the matrix cores, SiLU, attention gate (GCL) or coordinate head (EquivariantUpdate), neighbour
// sum - i.e. of egnn_new.py:35-56 / :91-104 as k_edge computes them.  Included through kernels.hpp.
//
// Per-edge activations are not kept by the forward pass; they are recomputed here tile by tile (32 edge rows per
// wavefront, same tile tables as the forward kernel) in two stages:
//   stage A  pre2 = W2 P + b2 (MFMA, as in the forward kernel), M = silu(pre2), gate / head, and from the incoming
//            gradient of the neighbour sum:  G2 = dL/d(pre2)  [E_pad][H]   (+ per-tile partials of d(wa), d(ba) / d(w7),
//            COORD: per-edge d(unit direction), d(phi))
//   stage B  dP = G2 W2 (MFMA with the transposed weight image), pre1 recomputed from the AB rows,
//            P = silu(pre1), G1 = dP * silu'(pre1) = dL/d(pre1)  [E_pad][H]   (+ per-edge d(radial), d(d0))
// followed by two deterministic CSR reductions (rows by receiving node -> dA, rows by sending node -> dB) and a per-node
// coordinate-gradient kernel.  The dense reduction over all edges dW2 = G2^T P runs on the library's own GEMMs (hd_gemm_f32 /
// hd_dw2_f16) over the materialised [E_pad][H] operands (hierdiff_amd/training.py).
// The weight chunks are double-buffered in LDS (one barrier per K chunk; the next chunk's stream and the next operand's
// row gathers are in flight under the current chunk's MFMAs) and the kernels are held to 256 registers, so two workgroups
// share a CU and one wavefront's epilogue runs under the other's matrix work (the fp32 MFMA hides nothing inside a wavefront,
// DESIGN.md section 4).  The column reductions that need no second pass over the materialised operands - db2 = colsum(G2),
// d(w_r), d(w_d) = {radial, d0}^T G1 - leave the kernels as per-tile partial sums (b2part, wrdpart).
#pragma once
#include "common.hpp"

struct EdgeBwdArgs {
    // forward inputs of the layer
    const float* AB;        // [M_pad][2H]
    const float* wrd;       // [2][H] w_r, w_d
    const float* Wimg;      // stage A: chunk image of W2;  stage B: chunk image of W2^T
    const float* b2;        // [H]
    const float* wa;        // [H] att_mlp.0.weight / coord_mlp.4.weight
    const int* ei;
    const int* ej;
    const uint8_t* eseg;
    const float* xcur;      // [M_pad][4]
    const float* x0;        // [M_pad][4]
    const float* ba_ptr;    // device pointer to the attention bias (NULL: 0)
    float norm_constant, coords_range, inv_norm;
    int attention, use_tanh, n_tiles;
    // stage A
    const float* pre2;      // SAVED: [tiles][H/32][4][64][4] second-layer pre-activations kept by the forward kernel (HD_EDGE_SAVE)
    const float* gin;       // GCL: d(agg) [M_pad][H];  COORD: d(xagg) [M_pad][4]
    float* G2;              // [E_pad][H]
    float* escal;           // [E_pad][8]: {du_x, du_y, du_z, dphi, d(radial), d(d0), -, -}
    float* colpart;         // [tiles][H] per-tile partial of d(wa) / d(w7)
    float* bapart;          // [tiles]    per-tile partial of d(ba)
    float* b2part;          // [tiles][H] per-tile column sums of G2 (d(b2))
    // stage B
    float* Pout;            // [E_pad][H]
    float* G1;              // [E_pad][H]
    float* wrdpart;         // [tiles][2][H] per-tile sum_r radial_r G1[r][:], sum_r d0_r G1[r][:]   (d(w_r), d(w_d))
    // fp16x3 contractions (training_precision "fp16x3"; only with SAVED stage A)
    float* g2max;           // [E_pad] max_c |G2[e][c]| - written by stage A (SAVED), ranges the rows of stage B's operand
    float* g2wgmax;         // [workgroups] the same maximum per workgroup of stage A  (global range of G2 for k_dw2_f16)
    float* pwgmax;          // [workgroups] max |P| per workgroup of stage B (PREC 3)   (global range of P for k_dw2_f16)
    const float* w2scal;    // stage B PREC 3: device scalars of the fp16 weight image {2^k, 2^-k, ...} (k_f16_prep)
};

// sigmoid and the SiLU derivative from it: silu'(x) = s (1 + x (1 - s))
HD_DEVINL float dsilu_from_sigmoid(float x, float s) { return s * __builtin_fmaf(x, 1.0f - s, 1.0f); }

// STAGE 0 = A, 1 = B.  One workgroup = four 32-row tiles (one per wavefront), grid = tiles / 4.
// PREC 0: the two H x H contractions (stage A: pre2 = W2 P, stage B: dP = G2 W2) in exact fp32.  (The PREC 2 branches - the retired
// `training_precision = "bf16x6"` of rounds 4-5: three-way bf16 split, six MFMAs per product - are compiled by no launch site since
// ABI 12; everything around a contraction - first-layer recomputation, SiLU and its derivative, gate / head, the materialised
// G2 / P / G1 tiles, per-tile partial sums - is the fp32 code of PREC 0 in every arithmetic.)
// SAVED (stage A only, round 5): the forward pass kept pre2 (k_edge with HD_EDGE_SAVE, 32 H floats per tile in accumulator order); the
// accumulators are loaded instead of recomputed - no weight stream, no MFMA, 32 16-byte loads per lane up front - and the stage is
// the HBM-bound element-wise kernel it is at heart (reads pre2 + the gathered gradient rows, writes G2: 2 x 4 H bytes per edge row).
// PREC 3 (stage B only, round 5, `training_precision = "fp16x3"`): dP = G2 W2 in the two-way FP16 split of the sampler's fp16x3 mode (three
// v_mfma_f32_32x32x16_f16 per product).  An operand row - the G2 row of one edge - is scaled by s = 2^(13 - E), E = floor(log2(max_c
// |G2[e][c]|)), the row maximum stage A left in g2max (exact, not a bound), the W2^T image by the power of two that puts its largest
// element into [2^14, 2^15) (k_pack_w2_f16); the epilogue undoes both in the factor it multiplies a row by anyway.
template <int H, bool COORD, int STAGE, int PREC = 0, bool SAVED = false>
__global__ __launch_bounds__(256, 2) void k_edge_bwd(EdgeBwdArgs a) {
    static_assert(!SAVED || STAGE == 0, "only stage A has something to load");
    static_assert(PREC != 3 || (STAGE == 1 && H >= 128), "the fp16x3 contraction exists for stage B at widths 128 / 256");
    constexpr int KC = (PREC == 2 || PREC == 3) ? 16 : 32;         // K chunk width
    constexpr int NCT = H / 32, NCH = H / KC, CHF = PREC == 2 ? 24 * H : PREC == 3 ? 16 * H : 32 * H;   // CHF: floats per weight chunk image
    extern __shared__ __attribute__((aligned(16))) float smem_b[];
    float* wbuf0 = smem_b;                       // [2][CHF] two K chunks of the weight image (double buffer)
    float* scr = smem_b + (SAVED ? 0 : 2 * CHF); // per wave: 32 phi + 96 unit dir + 32 ni + 32 nj + 32 radial + 32 d0 + 32 valid
    __shared__ __attribute__((aligned(16))) float wrd_s[4 * H];    // [w_r | w_d | b2 | wa]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    float* my = scr + wave * 288;
    float* phi_s = my;
    float* tr = my + 32;
    int* rowi_s = reinterpret_cast<int*>(my + 128);
    int* rowj_s = reinterpret_cast<int*>(my + 160);
    float* rrad_s = my + 192;
    float* rd0_s = my + 224;
    float* rval_s = my + 256;

    for (int k = tid; k < 2 * H; k += 256) wrd_s[k] = a.wrd[k];
    for (int k = tid; k < H; k += 256) { wrd_s[2 * H + k] = a.b2[k]; wrd_s[3 * H + k] = a.wa[k]; }

    // XCD-aware placement as in k_edge (round 5): block b runs on XCD b % 8 and takes the (b / 8)-th workgroup-tile of that XCD's
    // contiguous share of the edge list, so the AB / gradient rows of a molecule are gathered into ONE L2 (dealt round-robin, stage B
    // read 364 MB for 228 MB of G2: profiles/r05_pmc_train_hbm.log)
    int wt;
    {
        const int bid = blockIdx.x, nwt = gridDim.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nwt >> 3, r = nwt & 7;
        wt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;        // a bijection of [0, nwt): slot < q + (xcd < r) always
    }
    const int tile = wt * 4 + wave;                            // every tile of the (padded) table exists
    const int e = tile * 32 + n;
    const int ni = a.ei[e], nj = a.ej[e];
    const float valid = (a.eseg[e] != 255) ? 1.0f : 0.0f;
    const f32x4 xi = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)ni * 4);
    const f32x4 xj = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)nj * 4);
    const f32x4 yi = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)ni * 4);
    const f32x4 yj = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)nj * 4);
    const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
    const float radial = dx * dx + dy * dy + dz * dz;
    const float ex = yi[0] - yj[0], ey = yi[1] - yj[1], ez = yi[2] - yj[2];
    const float d0 = ex * ex + ey * ey + ez * ez;
    if (hh == 0) {
        rowi_s[n] = ni; rowj_s[n] = nj; rrad_s[n] = radial; rd0_s[n] = d0; rval_s[n] = valid;
        if constexpr (COORD) {
            const float inv = valid / (sqrtf(radial + 1e-8f) + a.norm_constant);
            tr[n * 3 + 0] = dx * inv; tr[n * 3 + 1] = dy * inv; tr[n * 3 + 2] = dz * inv;
        }
    }

    // fp16x3 stage B: scale of this lane's operand row and, for the epilogue (by row slot), 1 / (row scale x image scale)
    float f16_s = 1.0f;
    if constexpr (PREC == 3) {
        const float g = fmaxf(a.g2max[e], 7.8886090522101181e-31f);                   // 2^-100: a zero row stays zero under any scale
        const uint32_t eb = (__builtin_bit_cast(uint32_t, g) >> 23) & 0xffu;          // g in [2^(eb-127), 2^(eb-126))
        f16_s = __builtin_bit_cast(float, (267u - eb) << 23);                         // 2^(13 - E): g s < 2^14
        if (hh == 0) phi_s[n] = __builtin_bit_cast(float, (eb - 13u) << 23) * a.w2scal[1];
    }

    // A operand of K chunk c for this lane's edge row (k = 32c + 16hh + 0..15).  The raw rows are requested one chunk ahead
    // (load_raw, in flight under the MFMAs of the current chunk) and finished behind them (finish_P: first-layer
    // pre-activation + SiLU; stage B: the G2 row as it is).
    constexpr int NQ = KC / 8;                                     // float4 row pieces per lane and chunk (k = KC c + (KC/2) hh + 0 .. KC/2-1)
    const float* Arow = a.AB + (size_t)ni * (2 * H) + (KC / 2) * hh;
    const float* Brow = a.AB + (size_t)nj * (2 * H) + H + (KC / 2) * hh;
    const float* Grow = a.G2 + (size_t)e * H + (KC / 2) * hh;
    struct Raw { f32x4 a[NQ]; f32x4 b[NQ]; };
    auto load_raw = [&](int c, Raw& w) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            if constexpr (STAGE == 0) {
                w.a[u] = *reinterpret_cast<const f32x4*>(Arow + KC * c + 4 * u);
                w.b[u] = *reinterpret_cast<const f32x4*>(Brow + KC * c + 4 * u);
            } else {
                w.a[u] = *reinterpret_cast<const f32x4*>(Grow + KC * c + 4 * u);
            }
        }
    };
    auto finish_P = [&](int c, const Raw& w, float (&P)[4 * NQ]) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            if constexpr (STAGE == 0) {
                const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wrd_s + KC * c + (KC / 2) * hh + 4 * u);
                const f32x4 wd4 = *reinterpret_cast<const f32x4*>(wrd_s + H + KC * c + (KC / 2) * hh + 4 * u);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pre = w.a[u][j] + w.b[u][j];               // same operation order as the forward kernel
                    pre = __builtin_fmaf(radial, wr4[j], pre);
                    pre = __builtin_fmaf(d0, wd4[j], pre);
                    P[4 * u + j] = HD_F32_SILU(pre);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) P[4 * u + j] = PREC == 3 ? w.a[u][j] * f16_s : w.a[u][j];
            }
        }
    };
    auto issue_chunk = [&](int c) {                              // 1 KiB per wave-instruction, lane-linear image
        const float* src = a.Wimg + (size_t)c * CHF;
        float* dst = wbuf0 + (c & 1) * CHF;
        for (int k = tid * 4; k < CHF; k += 256 * 4) glds16(src + k, dst + (k - lane * 4));
    };

    f32x16 acc[NCT];
    if constexpr (SAVED) {
        const float* src = a.pre2 + (size_t)tile * (32 * H) + lane * 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (ct * 4 + q) * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ct][4 * q + j] = v[j];
            }
        __syncthreads();                                         // wrd_s staged
    } else {
    issue_chunk(0);
    __syncthreads();                                             // wrd_s staged
    float P[4 * NQ];
    Raw raw;
    load_raw(0, raw);
    finish_P(0, raw, P);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const float b2v = (STAGE == 0) ? wrd_s[2 * H + 32 * ct + n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = b2v;
    }
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of chunk c have landed
        __syncthreads();                                         // chunk c is complete; nobody reads the other buffer any more
        if (c + 1 < NCH) { issue_chunk(c + 1); load_raw(c + 1, raw); }
        const float* wbuf = wbuf0 + (c & 1) * CHF;
        if constexpr (PREC == 2) {
            // one k-step of 16 per chunk: the lane's 8 operand values -> head / middle / tail dwords (k order = element order of the
            // image: lane (hh, n) element i is W[32 ct + n][16 c + 8 hh + i]); per pair of column tiles two groups of six MFMAs on
            // alternating accumulators, the small terms first (h*L, h*M, m*M | h*H, m*H, l*H), fragments requested a pair ahead
            u32x4 xh, xm, xl;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint32_t hi[2], mi[2], lo[2];
                bf16_split3(P[4 * u], P[4 * u + 1], hi[0], mi[0], lo[0]);
                bf16_split3(P[4 * u + 2], P[4 * u + 3], hi[1], mi[1], lo[1]);
                xh[2 * u] = hi[0]; xh[2 * u + 1] = hi[1]; xm[2 * u] = mi[0]; xm[2 * u + 1] = mi[1]; xl[2 * u] = lo[0]; xl[2 * u + 1] = lo[1];
            }
            const bf16x8 A_h = __builtin_bit_cast(bf16x8, xh), A_m = __builtin_bit_cast(bf16x8, xm), A_l = __builtin_bit_cast(bf16x8, xl);
            const bf16x8* wf = reinterpret_cast<const bf16x8*>(wbuf) + lane;          // [piece][ct][64 lanes] x 16 B
            bf16x8 fcur[6], fnxt[6];
            auto load_f = [&](int g, bf16x8 (&f)[6]) {
#pragma unroll
                for (int p = 0; p < 3; ++p) { f[2 * p] = wf[(p * NCT + 2 * g) * 64]; f[2 * p + 1] = wf[(p * NCT + 2 * g + 1) * 64]; }
            };
            load_f(0, fcur);
#pragma unroll
            for (int g = 0; g < NCT / 2; ++g) {
                if (g + 1 < NCT / 2) load_f(g + 1, fnxt);
                const int c0 = 2 * g, c1 = 2 * g + 1;
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, fcur[4], acc[c0], 0, 0, 0);       // h * L
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, fcur[5], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, fcur[2], acc[c0], 0, 0, 0);       // h * M
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, fcur[3], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, fcur[2], acc[c0], 0, 0, 0);       // m * M
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, fcur[3], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, fcur[0], acc[c0], 0, 0, 0);       // h * H
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, fcur[1], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, fcur[0], acc[c0], 0, 0, 0);       // m * H
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, fcur[1], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_l, fcur[0], acc[c0], 0, 0, 0);       // l * H
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_l, fcur[1], acc[c1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 6; ++k) fcur[k] = fnxt[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (PREC == 3) {
            // one k-step of 16 per chunk; image [hi | lo][column tile][64 lanes] x 16 B (k_pack_w2_f16c).  Per pair of column tiles six
            // MFMAs on alternating accumulators (h*H, l*H, h*L), fragments requested a pair ahead.
            u32x4 xh, xl;
#pragma unroll
            for (int i = 0; i < 4; ++i) { uint32_t hi, lo; f16_split2(P[2 * i], P[2 * i + 1], hi, lo); xh[i] = hi; xl[i] = lo; }
            const f16x8 A_h = __builtin_bit_cast(f16x8, xh), A_l = __builtin_bit_cast(f16x8, xl);
            const f16x8* wf = reinterpret_cast<const f16x8*>(wbuf) + lane;          // [piece][ct][64 lanes] x 16 B
            f16x8 fcur[4], fnxt[4];
            auto load_f = [&](int g, f16x8 (&f)[4]) {
                f[0] = wf[(2 * g) * 64]; f[1] = wf[(NCT + 2 * g) * 64]; f[2] = wf[(2 * g + 1) * 64]; f[3] = wf[(NCT + 2 * g + 1) * 64];
            };
            load_f(0, fcur);
#pragma unroll
            for (int g = 0; g < NCT / 2; ++g) {
                if (g + 1 < NCT / 2) load_f(g + 1, fnxt);
                const int c0 = 2 * g, c1 = 2 * g + 1;
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, fcur[0], acc[c0], 0, 0, 0);       // h * H
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, fcur[2], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_l, fcur[0], acc[c0], 0, 0, 0);       // l * H
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_l, fcur[2], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, fcur[1], acc[c0], 0, 0, 0);       // h * L
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, fcur[3], acc[c1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) fcur[k] = fnxt[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // eight groups of (k-quad q, half of the column tiles); the B fragments of group g+1 are requested before the
            // MFMAs of group g.  sched_barrier keeps hipcc from hoisting more fragment reads than that (it spills otherwise).
            constexpr int GC = (NCT >= 2) ? NCT / 2 : 1, NG = 4 * (NCT / GC);
            f32x4 bcur[GC], bnxt[GC];
            auto load_b = [&](int g, f32x4 (&b)[GC]) {
                const int q = g / (NCT / GC), c0 = (g % (NCT / GC)) * GC;
    #pragma unroll
                for (int k = 0; k < GC; ++k) b[k] = *reinterpret_cast<const f32x4*>(wbuf + ((q * NCT + c0 + k) * 64 + lane) * 4);
            };
            load_b(0, bcur);
    #pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int q = g / (NCT / GC), c0 = (g % (NCT / GC)) * GC;
                if (g + 1 < NG) load_b(g + 1, bnxt);
    #pragma unroll
                for (int j = 0; j < 4; ++j)
    #pragma unroll
                    for (int k = 0; k < GC; ++k)
                        acc[c0 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(P[4 * q + j], bcur[k][j], acc[c0 + k], 0, 0, 0);
    #pragma unroll
                for (int k = 0; k < GC; ++k) bcur[k] = bnxt[k];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (c + 1 < NCH) finish_P(c + 1, raw, P);
        __builtin_amdgcn_sched_barrier(0);
    }
    }

    // acc[ct][r] = row rho(r) = (r&3) + 8*(r>>2) + 4*hh, column 32*ct + n.
    // Row dots: transpose-reduce over the 32 lanes of a half (see k_edge); lanes 2s, 2s+1 end up with the dot of row slot s.
    auto row_reduce = [&](const float (&dot)[16]) -> float {
        float v8[8], v4[4], v2[2];
        const bool b4 = n & 16, b3 = n & 8, b2_ = n & 4, b1 = n & 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) v8[k] = (b4 ? dot[k + 8] : dot[k]) + __shfl_xor(b4 ? dot[k] : dot[k + 8], 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) v4[k] = (b3 ? v8[k + 4] : v8[k]) + __shfl_xor(b3 ? v8[k] : v8[k + 4], 8);
#pragma unroll
        for (int k = 0; k < 2; ++k) v2[k] = (b2_ ? v4[k + 2] : v4[k]) + __shfl_xor(b2_ ? v4[k] : v4[k + 2], 4);
        float v = (b1 ? v2[1] : v2[0]) + __shfl_xor(b1 ? v2[0] : v2[1], 2);
        return v + __shfl_xor(v, 1);
    };
    const int my_slot = (n >> 1) & 15;
    const int my_rho = (my_slot & 3) + 8 * (my_slot >> 2) + 4 * hh;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    auto rho = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * hh; };
    // The epilogues walk the accumulators one column tile (or one row) at a time; what they gather from global memory for
    // step k+1 is requested before the arithmetic of step k and sched_barrier(0) closes every step, so at most two steps'
    // worth of gathered values are live (left alone hipcc hoists all 128-256 gathers to the top and spills).

    if constexpr (STAGE == 0) {
        float gm[16];                                   // SAVED: running max_c |G2| of this lane's 16 rows
#pragma unroll
        for (int r = 0; r < 16; ++r) gm[r] = 0.f;
        // per-tile partials of d(wa) / d(w7) and of d(b2): the two halves of the wavefront added, one row of H per tile
        auto store_partials = [&](int ct, float cs, float bs) {
            const float cst = cs + __shfl_xor(cs, 32), bst = bs + __shfl_xor(bs, 32);
            if (hh == 0) {
                a.colpart[(size_t)tile * H + 32 * ct + n] = cst;
                a.b2part[(size_t)tile * H + 32 * ct + n] = bst;
            }
        };
        if constexpr (!COORD) {
            // incoming gradient of the neighbour sum at (row's receiving node, column): zero for padding rows
            const float* gbase[16];
            unsigned vmask = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gbase[r] = a.gin + (size_t)rowi_s[rho(r)] * H + n;
                vmask |= (rval_s[rho(r)] != 0.0f) ? (1u << r) : 0u;
            }
            auto load_g = [&](int ct, float (&g)[16]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = gbase[r][32 * ct];
            };
            auto gval = [&](const float (&g)[16], int r) { return ((vmask >> r) & 1u) ? g[r] * a.inv_norm : 0.0f; };
            float dot[16], sd[16], gc[16], gn[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { dot[r] = 0.f; sd[r] = 0.f; }
            load_g(0, gc);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (ct + 1 < NCT) load_g(ct + 1, gn);
                const float wav = wrd_s[3 * H + 32 * ct + n];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m = HD_F32_SILU(acc[ct][r]);
                    dot[r] = __builtin_fmaf(m, wav, dot[r]);
                    sd[r] = __builtin_fmaf(m, gval(gc, r), sd[r]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) gc[r] = gn[r];
                __builtin_amdgcn_sched_barrier(0);
            }
            const float rowdot = row_reduce(dot), rowsd = row_reduce(sd);
            float att = 1.0f, q = 0.0f;
            if (a.attention) { att = sigmoid_f(rowdot + (a.ba_ptr ? *a.ba_ptr : 0.0f)); q = rowsd * att * (1.0f - att); }
            {   // d(ba) of this tile: every row's q sits in two lanes of its half
                float qs = ((n & 1) == 0) ? q * rval_s[my_rho] : 0.0f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) qs += __shfl_xor(qs, o);
                if (lane == 0) a.bapart[tile] = qs;
            }
            float attr[16], qr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                attr[r] = __shfl(att, (lane & 32) | (2 * r));
                const float qv = __shfl(q, (lane & 32) | (2 * r));
                qr[r] = ((vmask >> r) & 1u) ? qv : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
            load_g(0, gc);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (ct + 1 < NCT) load_g(ct + 1, gn);
                const float wav = wrd_s[3 * H + 32 * ct + n];
                float* g2p = a.G2 + ((size_t)tile * 32 + 4 * hh) * H + 32 * ct + n;
                float cs = 0.f, bs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = acc[ct][r];
                    asm volatile("" : "+v"(x));        // opaque: hipcc otherwise keeps all 128 sigmoids of the first pass alive for this one (spills)
                    const float s = sigmoid_f(x);
                    const float m = x * s;
                    const float dM = __builtin_fmaf(qr[r], wav, gval(gc, r) * attr[r]);     // d(msg)*att + (d(msg).M) att(1-att) wa
                    const float g2 = dM * dsilu_from_sigmoid(x, s);                          // zero for padding rows (g, qr)
                    g2p[(size_t)((r & 3) + 8 * (r >> 2)) * H] = g2;
                    if constexpr (SAVED) gm[r] = fmaxf(gm[r], fabsf(g2));
                    cs = __builtin_fmaf(qr[r], m, cs);
                    bs += g2;
                }
                store_partials(ct, cs, bs);
#pragma unroll
                for (int r = 0; r < 16; ++r) gc[r] = gn[r];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float dot[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dot[r] = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float wav = wrd_s[3 * H + 32 * ct + n];
#pragma unroll
                for (int r = 0; r < 16; ++r) dot[r] = __builtin_fmaf(HD_F32_SILU(acc[ct][r]), wav, dot[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float rowdot = row_reduce(dot);                     // phi of row rho(my_slot)
            if (lane == 0) a.bapart[tile] = 0.0f;                     // no attention bias in a coordinate layer (saves the host a fill)
            if ((n & 1) == 0) phi_s[my_rho] = rowdot;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (hh == 0) {                                              // lane n handles row n
                const float phi = phi_s[n];
                const float ux = tr[n * 3], uy = tr[n * 3 + 1], uz = tr[n * 3 + 2];   // zero for padding rows
                const float* gp = a.gin + (size_t)rowi_s[n] * 4;
                const float gx = gp[0] * a.inv_norm, gy = gp[1] * a.inv_norm, gz = gp[2] * a.inv_norm;
                const float t = tanhf(phi);
                const float sc = a.use_tanh ? t * a.coords_range : phi;              // trans = u * sc
                const float dsc = gx * ux + gy * uy + gz * uz;
                const float dphi = a.use_tanh ? dsc * a.coords_range * (1.0f - t * t) : dsc;
                const float v = rval_s[n];
                float* es = a.escal + ((size_t)tile * 32 + n) * 8;
                es[0] = v * gx * sc; es[1] = v * gy * sc; es[2] = v * gz * sc; es[3] = dphi;
                phi_s[n] = dphi;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float dphir[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dphir[r] = phi_s[rho(r)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float wav = wrd_s[3 * H + 32 * ct + n];
                float* g2p = a.G2 + ((size_t)tile * 32 + 4 * hh) * H + 32 * ct + n;
                float cs = 0.f, bs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = acc[ct][r];
                    asm volatile("" : "+v"(x));        // opaque, as in the GCL variant
                    const float s = sigmoid_f(x);
                    const float g2 = dphir[r] * wav * dsilu_from_sigmoid(x, s);     // dphi is zero for padding rows
                    g2p[(size_t)((r & 3) + 8 * (r >> 2)) * H] = g2;
                    if constexpr (SAVED) gm[r] = fmaxf(gm[r], fabsf(g2));
                    cs = __builtin_fmaf(dphir[r], x * s, cs);
                    bs += g2;
                }
                store_partials(ct, cs, bs);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (SAVED) {
            // row maxima of G2 (the range of stage B's fp16x3 operand rows): the transposed reduction of the row dots with max for +
            float v8[8], v4[4], v2[2];
            const bool b4 = n & 16, b3 = n & 8, b2_ = n & 4, b1 = n & 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = fmaxf(b4 ? gm[k + 8] : gm[k], __shfl_xor(b4 ? gm[k] : gm[k + 8], 16));
#pragma unroll
            for (int k = 0; k < 4; ++k) v4[k] = fmaxf(b3 ? v8[k + 4] : v8[k], __shfl_xor(b3 ? v8[k] : v8[k + 4], 8));
#pragma unroll
            for (int k = 0; k < 2; ++k) v2[k] = fmaxf(b2_ ? v4[k + 2] : v4[k], __shfl_xor(b2_ ? v4[k] : v4[k + 2], 4));
            float v = fmaxf(b1 ? v2[1] : v2[0], __shfl_xor(b1 ? v2[0] : v2[1], 2));
            v = fmaxf(v, __shfl_xor(v, 1));
            if (a.g2max) {
                if ((n & 1) == 0) a.g2max[(size_t)tile * 32 + my_rho] = v;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
                __shared__ float wgm[4];
                if (lane == 0) wgm[wave] = v;
                __syncthreads();
                if (tid == 0) a.g2wgmax[blockIdx.x] = fmaxf(fmaxf(wgm[0], wgm[1]), fmaxf(wgm[2], wgm[3]));
            }
        }
    } else {
        // stage B: row by row (the two AB rows of an edge are gathered once per row, one row ahead); a row's two dots with
        // w_r / w_d are reduced over its half-wave on the spot (keeping 2 x 16 running dots for a transposed reduction at
        // the end costs the registers that make the difference between one and two wavefronts per SIMD)
        float wrp[NCT], wdp[NCT];
        float pmx = 0.f;                                            // PREC 3: max |P| over this lane's share of the tile
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { wrp[ct] = 0.f; wdp[ct] = 0.f; }
        auto load_ab = [&](int r, float (&ai)[NCT], float (&bj)[NCT]) {
            const float* pa = a.AB + (size_t)rowi_s[rho(r)] * (2 * H) + n;
            const float* pb = a.AB + (size_t)rowj_s[rho(r)] * (2 * H) + H + n;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) { ai[ct] = pa[32 * ct]; bj[ct] = pb[32 * ct]; }
        };
        float aic[NCT], bjc[NCT], ain[NCT], bjn[NCT];
        load_ab(0, aic, bjc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r + 1 < 16) load_ab(r + 1, ain, bjn);
            const float rad = rrad_s[rho(r)], dd0 = rd0_s[rho(r)], vrr = rval_s[rho(r)];
            const float vrs = PREC == 3 ? vrr * phi_s[rho(r)] : vrr;       // fp16x3: the accumulators carry row scale x image scale
            float* po = a.Pout + ((size_t)tile * 32 + rho(r)) * H + n;
            float* go = a.G1 + ((size_t)tile * 32 + rho(r)) * H + n;
            float dr = 0.f, dd = 0.f, pmr = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int col = 32 * ct + n;
                const float wr = wrd_s[col], wd = wrd_s[H + col];
                float pre = aic[ct] + bjc[ct];
                pre = __builtin_fmaf(rad, wr, pre);
                pre = __builtin_fmaf(dd0, wd, pre);
                const float s = sigmoid_f(pre);
                const float g1 = vrs * acc[ct][r] * dsilu_from_sigmoid(pre, s);
                const float pv = vrr * pre * s;
                po[32 * ct] = pv;
                // (opaque: as plain fmaxf hipcc re-associates the maximum over the whole tile into a tree and spills 83 registers)
                if constexpr (PREC == 3) asm("v_max_f32 %0, %0, |%1|" : "+v"(pmr) : "v"(pv));
                go[32 * ct] = g1;
                dr = __builtin_fmaf(g1, wr, dr);
                dd = __builtin_fmaf(g1, wd, dd);
                wrp[ct] = __builtin_fmaf(g1, rad, wrp[ct]);
                wdp[ct] = __builtin_fmaf(g1, dd0, wdp[ct]);
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { dr += __shfl_xor(dr, o); dd += __shfl_xor(dd, o); }
            if constexpr (PREC == 3) pmx = fmaxf(pmx, pmr);
            if (n == 0) {                                            // d(radial), d(d0) of row rho(r)
                float* es = a.escal + ((size_t)tile * 32 + rho(r)) * 8;
                es[4] = dr; es[5] = dd;
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) { aic[ct] = ain[ct]; bjc[ct] = bjn[ct]; }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {                           // per-tile partials of d(w_r), d(w_d)
            const float wrt = wrp[ct] + __shfl_xor(wrp[ct], 32), wdt = wdp[ct] + __shfl_xor(wdp[ct], 32);
            if (hh == 0) {
                a.wrdpart[((size_t)tile * 2) * H + 32 * ct + n] = wrt;
                a.wrdpart[((size_t)tile * 2 + 1) * H + 32 * ct + n] = wdt;
            }
        }
        if constexpr (PREC == 3) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) pmx = fmaxf(pmx, __shfl_xor(pmx, o));
            __shared__ float wpm[4];
            if (lane == 0) wpm[wave] = pmx;
            __syncthreads();
            if (tid == 0) a.pwgmax[blockIdx.x] = fmaxf(fmaxf(wpm[0], wpm[1]), fmaxf(wpm[2], wpm[3]));
        }
    }
}

// ----------------------------------------------------------------------------- reductions after the two stages

// out[i][col0 + c] = sum over the rows listed for node i (CSR, ascending row order => deterministic) of G[row][c]
struct CsrSumArgs {
    const float* G;         // [E_pad][H]
    const int* ptr;         // [M+1]
    const int* rows;        // [E]
    float* out;             // [M_pad][ldo]
    int M, H, ldo, col0;
    const float* G2;        // optional second operand [E_pad][4] (stage 2: the per-edge translations), summed by one more
    float* out2;            // thread per node into [M_pad][4]; NULL = not present
    const float* x_in;      // optional (with G2; stage 2, round 5): x_out[i][k] = (x_in[i][k] + sum[k]) * xmask[i], k < 3 -
    float* x_out;           // k_egcl_node_out's coordinate expression, saving that launch
    const float* xmask;     // [M] or NULL
    const int* ptr_b;       // gridDim.y == 2 (the edge layer's backward): blockIdx.y = 1 sums the rows of a second CSR (rows by
    const int* rows_b;      // sending node) into columns col0_b .. of the same output - both reductions of G1 in one launch
    int col0_b;
};

__global__ void k_csr_sum(CsrSumArgs a) {
    const int q = a.H >> 2, qt = q + (a.G2 ? 1 : 0);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = idx / qt, c4 = idx - i * qt;
    if (i >= a.M) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    // rows are added in ascending list order (deterministic; the order every caller's parity rests on), but REQUESTED four at a
    // time: written as one load per iteration the loop is a chain of dependent L2 round trips (index -> row) per edge - 12.3 us
    // for the 12 incoming edges of a stage-2 node (round 5)
    const bool second = blockIdx.y == 1;
    const int* const ptr = second ? a.ptr_b : a.ptr;
    const int* const rows = second ? a.rows_b : a.rows;
    const int col0 = second ? a.col0_b : a.col0;
    const int p0 = ptr[i], p1 = ptr[i + 1];
    const float* base = (c4 == q) ? a.G2 : a.G + 4 * c4;
    const size_t ld = (c4 == q) ? 4 : (size_t)a.H;
    int p = p0;
    for (; p + 4 <= p1; p += 4) {
        const int r0 = rows[p], r1 = rows[p + 1], r2 = rows[p + 2], r3 = rows[p + 3];
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(base + (size_t)r0 * ld), g1 = *reinterpret_cast<const f32x4*>(base + (size_t)r1 * ld);
        const f32x4 g2 = *reinterpret_cast<const f32x4*>(base + (size_t)r2 * ld), g3 = *reinterpret_cast<const f32x4*>(base + (size_t)r3 * ld);
        v += g0; v += g1; v += g2; v += g3;
    }
    for (; p < p1; ++p) v += *reinterpret_cast<const f32x4*>(base + (size_t)rows[p] * ld);
    if (c4 == q) {
        *reinterpret_cast<f32x4*>(a.out2 + (size_t)i * 4) = v;
        if (a.x_out) {
            const float m = a.xmask ? a.xmask[i] : 1.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) a.x_out[(size_t)i * 3 + k] = (a.x_in[(size_t)i * 3 + k] + v[k]) * m;
        }
    } else *reinterpret_cast<f32x4*>(a.out + (size_t)i * a.ldo + col0 + 4 * c4) = v;
}

// Coordinate gradients of one edge layer from the per-edge scalars: radial = |x_i - x_j|^2 (d(radial) from stage B), the
// same for d0 on the input coordinates, and - coordinate layers - the unit direction u = diff / (sqrt(radial + 1e-8) + nc).
struct EdgeDxArgs {
    const float* escal;     // [E_pad][8]
    const int* ei;
    const int* ej;
    const int* rptr; const int* rrows;      // rows by receiving node
    const int* sptr; const int* srows;      // rows by sending node
    const float* xcur; const float* x0;     // [M_pad][4]
    float* dx; float* dx0;                  // [M_pad][4] (overwritten)
    float norm_constant;
    int M, coord;
};

// One WAVEFRONT per node (round 5; was one thread per node: 7,680 threads each walking 58 edges through a three-deep chain of
// dependent loads - 67 us, 18 times per training step).  Lane l takes the node's l-th incident edge - its receiving edges first,
// then its sending ones - in rounds of 64; the six partial sums are added across the lanes by a fixed butterfly and across the
// rounds in order: deterministic, though not the sequential order of the old kernel (gradients agree to round-off).
__global__ void k_edge_dx(EdgeDxArgs a) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= a.M) return;
    const f32x4 xk = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)k * 4);
    const f32x4 yk = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)k * 4);
    const int r0 = a.rptr[k], nr = a.rptr[k + 1] - r0, s0 = a.sptr[k], ns = a.sptr[k + 1] - s0;
    float tx = 0.f, ty = 0.f, tz = 0.f, ux = 0.f, uy = 0.f, uz = 0.f;
    for (int base = 0; base < nr + ns; base += 64) {
        const int l = base + lane;
        float gx = 0.f, gy = 0.f, gz = 0.f, hx = 0.f, hy = 0.f, hz = 0.f;
        if (l < nr + ns) {
            const bool recv = l < nr;
            const int row = recv ? a.rrows[r0 + l] : a.srows[s0 + (l - nr)];
            const int other = recv ? a.ej[row] : a.ei[row];
            const float sign = recv ? 1.0f : -1.0f;
            // diff = x_recv - x_send; this node is the receiver (sign +1) or the sender (sign -1)
            const f32x4 xo = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)other * 4);
            const f32x4 yo = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)other * 4);
            const f32x4 e03 = *reinterpret_cast<const f32x4*>(a.escal + (size_t)row * 8);
            const f32x4 e47 = *reinterpret_cast<const f32x4*>(a.escal + (size_t)row * 8 + 4);
            const float ddx = sign * (xk[0] - xo[0]), ddy = sign * (xk[1] - xo[1]), ddz = sign * (xk[2] - xo[2]);   // = diff
            float cx = 2.0f * e47[0] * ddx, cy = 2.0f * e47[0] * ddy, cz = 2.0f * e47[0] * ddz;                      // via radial
            if (a.coord) {
                const float rad = ddx * ddx + ddy * ddy + ddz * ddz;
                const float sq = sqrtf(rad + 1e-8f), nrm = sq + a.norm_constant;
                const float dot = e03[0] * ddx + e03[1] * ddy + e03[2] * ddz;
                const float k2 = dot / (nrm * nrm * sq);
                cx += e03[0] / nrm - k2 * ddx; cy += e03[1] / nrm - k2 * ddy; cz += e03[2] / nrm - k2 * ddz;
            }
            gx = sign * cx; gy = sign * cy; gz = sign * cz;
            const float e0x = sign * (yk[0] - yo[0]), e0y = sign * (yk[1] - yo[1]), e0z = sign * (yk[2] - yo[2]);
            hx = sign * 2.0f * e47[1] * e0x; hy = sign * 2.0f * e47[1] * e0y; hz = sign * 2.0f * e47[1] * e0z;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o); gz += __shfl_xor(gz, o);
            hx += __shfl_xor(hx, o); hy += __shfl_xor(hy, o); hz += __shfl_xor(hz, o);
        }
        tx += gx; ty += gy; tz += gz; ux += hx; uy += hy; uz += hz;
    }
    if (lane == 0) {
        *reinterpret_cast<f32x4*>(a.dx + (size_t)k * 4) = f32x4{tx, ty, tz, 0.f};
        *reinterpret_cast<f32x4*>(a.dx0 + (size_t)k * 4) = f32x4{ux, uy, uz, 0.f};
    }
}

// W [H][H] (state_dict layout, row = output) -> chunk image of the forward fp32 edge kernel (pack_edge_w2), either of W
// (B operand W^T: image value W[col][k]) or of W^T (TRANS: image value W[k][col]); one thread per image float.
template <bool TRANS>
__global__ void k_pack_w2(const float* W, float* img, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * H) return;
    const int NCT = H / 32;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    const int rest = idx >> 8;                   // (c * 4 + q) * NCT + ct
    const int ct = rest % NCT, cq = rest / NCT, q = cq & 3, c = cq >> 2;
    const int k = 32 * c + 16 * (lane >> 5) + 4 * q + j, col = 32 * ct + (lane & 31);
    img[idx] = TRANS ? W[(size_t)k * H + col] : W[(size_t)col * H + k];
}
// both images of one W in one launch (the recomputing backward needs W2 for stage A and W2^T for stage B)
__global__ void k_pack_w2_both(const float* W, float* img, float* timg, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * H) return;
    const int NCT = H / 32;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    const int rest = idx >> 8;
    const int ct = rest % NCT, cq = rest / NCT, q = cq & 3, c = cq >> 2;
    const int k = 32 * c + 16 * (lane >> 5) + 4 * q + j, col = 32 * ct + (lane & 31);
    img[idx] = W[(size_t)col * H + k];
    timg[idx] = W[(size_t)k * H + col];
}
