// EXPERIMENT (round 2, not in the library): the fp32 node update of small batches as ONE launch on 16-row tiles and 16 x 16
// accumulators, meant to replace the k_agg + 3 x k_gemm chain (36-46 us per update) below 6,144 active rows.
// Result (profiles/history/r02_r16_sweep.log, scratch/r16_sweep.sh at the commit that carried it): BIT-IDENTICAL to k_node_f32 and
// to the chain (59 bitwise / parity tests green) and SLOWER at every size - forward B=2 0.83 -> 1.29 ms, B=64 1.96 -> 2.35,
// B=192 4.92 -> 5.69 - i.e. ~75 us per update.  Why: a workgroup pulls 1.3-1.8 MB of fp32 weights through ONE CU's L2 port;
// with three 32-wide K chunks in flight per wavefront (96 KB per CU) against a >= 2 us path (the 24 MB of weights do not
// stay in a 4 MB L2) Little's law allows about a third of the port's 61 B/clk, and the four phases run back to back.
// The 32-row k_node_f32 amortises the same stream over twice the rows, the GEMM chain spreads it over 120-480 workgroups.
// What the kernel would need is the weight stream staged through LDS by LDS-DMA (k_edge's scheme) or shared by a cluster
// of workgroups.  Kept for its verified 16 x 16 x 4 operand mapping (Mma16).
// To try it again: append to csrc/k_node.hpp, launch from launch_node_hw for mode 0 / upd / small M with
// grid (M + 15) / 16, block 64 * (H / 32), and produce the first layer's AB with the plain GEMM.
#pragma once

// ----------------------------------------------------------------------------- fused node update, fp32, 16-row tiles (small batches)
// k_node_f32's launch (neighbour-sum reduction, node MLP, residual, the next layers' first edge Linear) on 16-row tiles and
// v_mfma_f32_16x16x4_f32, for row counts that leave k_node_f32's 32-row workgroups (37 us of fp32 MFMA each) on a fraction
// of the chip: half the matrix work per workgroup, twice the workgroups.  One wavefront per 32-column tile of an H-wide
// output = two 16 x 16 accumulators sharing the A operand (LDS, fp32, k_node_f32's strides); weights L2 -> registers in
// k_node_f32's fragment images - fragment (k chunk s, column tile ct, q) is [64 lanes][4 j] with k = 32 s + 16 (lane >> 5)
// + 4 q + j, col = 32 ct + (lane & 31); the 16 x 16 tile t of a column tile reads lane' = 32 hf + 16 t + m and feeds
// instruction p = 2 q + pp with element 2 pp + (g >> 1): the k order of the 32 x 32 x 2 kernels, hence their bits
// (scratch/mb/mfma_order.hip).  Accumulators from zero, bias after the contraction, the same epilogue expressions as
// k_node_f32 / k_gemm, neighbour sums in part order: bit-identical to both (test_fp32_node_paths_agree_bitwise).
template <int KS, int NTW, int RD>
struct Mma16 {
    typedef f32x4 Ring[RD][NTW][2][4];
    // Wl: image + lane' base (32 hf + m); fragment (s, ct, q) of tile t at Wl[((s * NCTW + ct) * 4 + q) * 64 + 16 t]
    template <int s, int slot>
    static HD_DEVINL void load(Ring& br, const f32x4* Wl, const int (&ct)[NTW], int NCTW) {
#pragma unroll
        for (int c = 0; c < NTW; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) br[slot][c][t][q] = Wl[((size_t)(s * NCTW + ct[c]) * 4 + q) * 64 + 16 * t];
    }
    static HD_DEVINL void prefetch(Ring& br, const f32x4* Wl, const int (&ct)[NTW], int NCTW) {
        static_for<0, (RD < KS ? RD : KS)>([&](auto S) { load<decltype(S)::value, decltype(S)::value>(br, Wl, ct, NCTW); });
    }
    // Arow: this lane's A row in LDS + 16 hf; odd = g >> 1
    static HD_DEVINL void run(f32x4 (&acc)[NTW][2], Ring& br, const float* Arow, const f32x4* Wl, const int (&ct)[NTW], int NCTW, int odd) {
        static_for<0, KS>([&](auto S) {
            constexpr int s = decltype(S)::value, slot = s % RD;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(Arow + 32 * s + 4 * q);
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const float a = odd ? av[2 * pp + 1] : av[2 * pp];
#pragma unroll
                    for (int c = 0; c < NTW; ++c)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const float b = odd ? br[slot][c][t][q][2 * pp + 1] : br[slot][c][t][q][2 * pp];
                            acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c][t], 0, 0, 0);
                        }
                }
            }
            if constexpr (s + RD < KS) load<s + RD, slot>(br, Wl, ct, NCTW);
        });
    }
};

template <int H, int NAB>
__global__ __launch_bounds__(64 * (H / 32)) void k_node_f32_r16(NodeArgs a) {
    constexpr int NW = H / 32, NT = 64 * NW, NCT = H / 32;
    constexpr int LDX = 2 * H + 4, LDH = H + 4;
    __shared__ __attribute__((aligned(16))) float X[16 * LDX];          // [h | agg], later h' in the h half
    __shared__ __attribute__((aligned(16))) float T[16 * LDH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, gs = lane >> 4, odd = gs >> 1, hf = gs & 1;
    const int row0 = blockIdx.x * 16;
    const int ct1[1] = {wave};
    const int ct2[2] = {wave, wave + NCT};
    typedef Mma16<2 * H / 32, 1, 3> M1;
    typedef Mma16<H / 32, 1, 3> M2;
    typedef Mma16<H / 32, 2, 2> M3;
    typename M1::Ring br1;
    typename M2::Ring br2;
    typename M3::Ring br3;
    const f32x4* W3l = reinterpret_cast<const f32x4*>(a.W3img) + 32 * hf + m;
    const f32x4* W4l = reinterpret_cast<const f32x4*>(a.W4img) + 32 * hf + m;
    M1::prefetch(br1, W3l, ct1, NCT);

    // ---- phase 0: X = [h | sum of parts / norm] -> LDS (k_node_f32's arithmetic: parts ascending, then / norm)
    {
        constexpr int Q = H / 4, TPR = NT / 16, NPT = Q / TPR;
        static_assert(Q % TPR == 0 && NPT >= 1, "row pieces must divide evenly");
        const int r = tid / TPR, cq = tid % TPR;
        const int row = row0 + r;
        int p0 = 0, p1 = 0;
        if (row < a.M) { p0 = a.pstart[row]; p1 = a.pstart[row + 1]; }
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const int c4 = 4 * (cq + u * TPR);
            *reinterpret_cast<f32x4*>(X + r * LDX + c4) = *reinterpret_cast<const f32x4*>(a.h_in + (size_t)row * H + c4);   // pad rows are zero
            f32x4 v = z4;
            for (int p = p0; p < p1; ++p) v += *reinterpret_cast<const f32x4*>(a.part + (size_t)p * H + c4);
            *reinterpret_cast<f32x4*>(X + r * LDX + H + c4) = v / a.norm;
        }
    }
    __syncthreads();
    // acc[c][t][i] = row 4 gs + i, column 32 ct[c] + 16 t + m
    // ---- phase 1: T = silu(X W3^T + b3)
    {
        f32x4 acc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
        M2::prefetch(br2, W4l, ct1, NCT);
        M1::run(acc, br1, X + m * LDX + 16 * hf, W3l, ct1, NCT, odd);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 32 * wave + 16 * t + m;
            const float b = a.b3[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) T[(4 * gs + i) * LDH + col] = silu_f(acc[0][t][i] + b);
        }
    }
    __syncthreads();
    // ---- phase 2: h' = (h + T W4^T + b4) * mask  -> X's h half and global
    const f32x4* AB0l = reinterpret_cast<const f32x4*>(a.ABimg[0]) + 32 * hf + m;
    {
        f32x4 acc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
        M3::prefetch(br3, AB0l, ct2, 2 * NCT);
        M2::run(acc, br2, T + m * LDH + 16 * hf, W4l, ct1, NCT, odd);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = 32 * wave + 16 * t + m;
            const float b = a.b4[col];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = 4 * gs + i;
                const float hres = X[R * LDX + col];               // = h_in[row0 + R][col]; only this lane touches the element
                const float v = (hres + (acc[0][t][i] + b)) * a.nmask[row0 + R];
                X[R * LDX + col] = v;
                if (row0 + R < a.M) a.h_out[(size_t)(row0 + R) * H + col] = v;
            }
        }
    }
    __syncthreads();
    // ---- phase 3: AB_q = h' [W1a | W1b]^T + bias, this wavefront's column tile of both H-wide halves
#pragma unroll
    for (int q = 0; q < NAB; ++q) {
        const f32x4* ABl = reinterpret_cast<const f32x4*>(a.ABimg[q]) + 32 * hf + m;
        if (q > 0) M3::prefetch(br3, ABl, ct2, 2 * NCT);
        f32x4 acc[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        M3::run(acc, br3, X + m * LDX + 16 * hf, ABl, ct2, 2 * NCT, odd);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = half * H + 32 * wave + 16 * t + m;
                const float b = a.ABbias[q][col];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int R = 4 * gs + i;
                    if (row0 + R < a.M) a.ABout[q][(size_t)(row0 + R) * 2 * H + col] = acc[half][t][i] + b;
                }
            }
    }
}
