// Node-side GEMM of the exact-fp32 sampler for batches below HD_FUSE_MIN_ROWS active rows (k_gemm_r16): the three
// contractions of a node update - T = silu([h | agg] W3^T + b3), h' = (h + T W4^T + b4) mask, AB_q = h' [W1a|W1b]_q^T + b -
// as three launches whose workgroups never wait for each other.  Included through kernels.hpp.
//
// Why not one fused launch (k_node_f32) here: a 32-row workgroup of the fused kernel is a 50 us serial chain (1.5 - 2 MB of
// weights and 37 us of fp32 MFMA through ONE CU), which 240 workgroups hide at B = 256 and 2 - 60 do not.  Why not an
// in-launch hand-off between column-split workgroups: an agent-scope release / acquire pair costs more than the 1.5 us of a
// kernel boundary on this chip (MI355X_MICROARCH.md, rows `boundary`, `barrier-xcd`, `splitk-seam`).  What the old chain
// (round 2: k_agg + 3 x k_gemm, or k_gemm_direct below 512 rows) paid for was per-chunk latency: 16 K chunks, each behind a barrier
// and an LDS round trip (k_gemm, 14 us for K = 512 whatever M) or behind a four-deep register ring of A AND B loads
// (k_gemm_direct).  Here:
//   * workgroup = 16 rows x 128 columns, eight wavefronts of one 16 x 16 accumulator each (v_mfma_f32_16x16x4_f32: a K = 512
//     chain is 128 dependent instructions = 4.1 k cycles instead of the 16.4 k of a 32 x 32 x 2 chain) - four times the
//     workgroups of a 64 x 64 tiling, so M = 1,920 rows (B = 64) fill the chip and M = 60 (B = 2) still spread over 8 - 32 CUs;
//   * the whole 16 x K A tile goes to LDS ONCE (one round of loads, one barrier), the neighbour-sum reduction of k_agg is
//     folded into that load (AGG: columns K1 .. K-1 are sum(parts) / norm, same order of additions, one launch less);
//   * weights go L2 -> registers per wavefront in fragment order, RING chunks ahead, no barrier in the K loop;
//   * two weight images can share a launch (the AB tables of the coordinate layer and of the next block's first GCL both
//     depend on the same h): six launches less per forward.
// Bit-identical to k_gemm and the fused k_node_f32: the 16 x 16 x 4 instruction fed the k values in the
// production order (v, v + 16 for v = 0 .. 15 per 32-wide chunk) is the same fmaf chain per output element
// (scratch/mb/mfma_order.hip), accumulators start at zero, bias / SiLU / residual-and-mask are k_gemm's epilogue expressions.
// Weight image (pack_gemm_b16): [16-column tile][32-wide K chunk][64 lanes][8 floats]; lane = (m, slot gs = (odd, hf)),
// value p is W[16 tile + m][32 chunk + 16 hf + 2 p + odd] - the k the lane feeds to instruction p of the chunk.
#pragma once
#include "k_node.hpp"

struct R16Args {
    const float* A;         // [M_pad][lda] columns k < K1
    const float* part;      // AGG: [P][K - K1] partial neighbour sums; pstart [M + 1]
    const int* pstart;
    const float* Bimg[2];   // weight images (the second only when n_img == 2)
    const float* bias[2];   // [Nc]
    float* C[2];            // [M_pad][ldc]
    const float* nmask;     // [M_pad] (EPI_RESID_MASK)
    float norm;
    int lda, ldc, K1, K, M, Nc, n_img;
};

// RT = 16-row tiles per workgroup.  The library launches RT = 1.  RT = 2 (the weight fragments a wavefront has pulled from L2
// serve two accumulators: at M = 1,920 rows the 16-row form re-reads every weight image 120 times, 61 MB per launch) was
// measured: slower up to B = 64 (two interleaved chains double the MFMA time per wavefront, the pipe not the L2 is the
// limit), equal from B = 128 (profiles/r03_r16_sweep2.log).
template <int EPI, bool AGG, int RT>
__global__ __launch_bounds__(512, 4) void k_gemm_r16(R16Args g) {
    constexpr int RING = 8;                                   // K chunks of weights in flight per wavefront (64 registers)
    extern __shared__ __attribute__((aligned(16))) float As[];   // [16 RT][K + 4], k permuted inside every 16-group (see below)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, gs = lane >> 4, odd = gs >> 1, hf = gs & 1;
    const int wpt = g.Nc >= 128 ? 8 : (g.Nc >> 4);           // wavefronts with a column tile (narrow layers: Nc = 32, 64)
    const int nct = g.Nc / (16 * wpt);                        // column tiles of 16 wpt columns per image
    const int per_rt = nct * g.n_img;
    const int rt = blockIdx.x / per_rt, rest = blockIdx.x - rt * per_rt;
    const int img = rest / nct, ctile = rest - img * nct;
    const int row0 = 16 * RT * rt;
    const int nchunk = g.K >> 5;
    const int LDA_S = g.K + 4;
    const int ct16 = wpt * ctile + (wave < wpt ? wave : 0);   // this wavefront's 16-column tile
    const f32x4* Bsrc = reinterpret_cast<const f32x4*>(g.Bimg[img]) + ((size_t)ct16 * nchunk * 64 + lane) * 2;
    const float bias = g.bias[img][16 * ct16 + m];             // requested here, used in the epilogue
    f32x4 rb[RING][2];
    // half of the ring goes out before the A tile's loads, the rest behind them (AGG: the fill below needs the registers)
    constexpr int PRE = AGG ? RING / 2 : RING;
    static_for<0, PRE>([&](auto Rc) {
        constexpr int r = decltype(Rc)::value;
        if (r < nchunk) { rb[r][0] = Bsrc[(size_t)r * 128]; rb[r][1] = Bsrc[(size_t)r * 128 + 1]; }
    });
    // ---- A tile -> LDS.  Position of original column o = 16 G + x inside its 16-group: 8 (x & 1) + (x >> 1), so that the
    // eight values a lane feeds to the eight instructions of a chunk (x = 2 p + odd) are contiguous: two ds_read_b128.
    // All global loads of a thread go out before the first is consumed (four float4 per thread at K = 512; a plain loop
    // costs one L2 round trip per piece - and two per neighbour-sum piece: pstart, then the parts - 12 instead of 6.5 us).
    {
        const int q4 = g.K >> 2;                              // float4 per row
        const int total = 16 * RT * q4;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        for (int base = 0; base < total; base += 4 * 512) {
            f32x4 v[4], g1[4];
            int pa[4], pb[4], rr[4], kk[4];
            bool ok[4], ag[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + tid + 512 * u;
                ok[u] = idx < total;
                const int r = idx / q4, c4 = idx - r * q4;
                rr[u] = r; kk[u] = 4 * c4;
                ag[u] = AGG && kk[u] >= g.K1;
                pa[u] = pb[u] = 0;
                v[u] = z4;
                if (ok[u] && !ag[u]) v[u] = *reinterpret_cast<const f32x4*>(g.A + (size_t)(row0 + r) * g.lda + kk[u]);   // pad rows are zero
                if (ok[u] && ag[u] && row0 + r < g.M) { pa[u] = g.pstart[row0 + r]; pb[u] = g.pstart[row0 + r + 1]; }
            }
            if constexpr (AGG) {
                const int HW = g.K - g.K1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {                 // the first two parts of every piece in flight together
                    g1[u] = z4;
                    if (ag[u] && pa[u] < pb[u]) v[u] = *reinterpret_cast<const f32x4*>(g.part + (size_t)pa[u] * HW + (kk[u] - g.K1));
                    if (ag[u] && pa[u] + 1 < pb[u]) g1[u] = *reinterpret_cast<const f32x4*>(g.part + (size_t)(pa[u] + 1) * HW + (kk[u] - g.K1));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ag[u]) {                              // k_agg: 0 + parts ascending, then / norm
                        f32x4 sacc = z4;
                        if (pa[u] < pb[u]) sacc += v[u];
                        if (pa[u] + 1 < pb[u]) sacc += g1[u];
                        for (int p = pa[u] + 2; p < pb[u]; ++p) sacc += *reinterpret_cast<const f32x4*>(g.part + (size_t)p * HW + (kk[u] - g.K1));
                        v[u] = sacc / g.norm;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ok[u]) {
                    float* dst = As + rr[u] * LDA_S + (kk[u] & ~15);
                    const int x0 = kk[u] & 15;                // 0, 4, 8, 12: x0 .. x0 + 3 -> positions (x0 >> 1) + {0, 8, 1, 9}
                    dst[(x0 >> 1)] = v[u][0]; dst[8 + (x0 >> 1)] = v[u][1]; dst[(x0 >> 1) + 1] = v[u][2]; dst[8 + (x0 >> 1) + 1] = v[u][3];
                }
            }
        }
    }
    static_for<PRE, RING>([&](auto Rc) {
        constexpr int r = decltype(Rc)::value;
        if (r < nchunk) { rb[r][0] = Bsrc[(size_t)r * 128]; rb[r][1] = Bsrc[(size_t)r * 128 + 1]; }
    });
    __syncthreads();
    if (wave >= wpt) return;
    const float* Arow = As + m * LDA_S + 16 * hf + 8 * odd;
    f32x4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < nchunk; c0 += RING) {
        static_for<0, RING>([&](auto Rc) {
            constexpr int r = decltype(Rc)::value;
            const int c = c0 + r;
            if (c < nchunk) {
                f32x4 a0[RT], a1[RT];
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    a0[t] = *reinterpret_cast<const f32x4*>(Arow + t * 16 * LDA_S + 32 * c);
                    a1[t] = *reinterpret_cast<const f32x4*>(Arow + t * 16 * LDA_S + 32 * c + 4);
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t][p], rb[r][0][p], acc[t], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][p], rb[r][1][p], acc[t], 0, 0, 0);
                if (c + RING < nchunk) { rb[r][0] = Bsrc[(size_t)(c + RING) * 128]; rb[r][1] = Bsrc[(size_t)(c + RING) * 128 + 1]; }
            }
        });
    }
    // acc[t][i] = row row0 + 16 t + 4 gs + i, column 16 ct16 + m; k_gemm's epilogue, element by element
    const int col = 16 * ct16 + m;
    float* Cc = g.C[img];
#pragma unroll
    for (int ti = 0; ti < 4 * RT; ++ti) {
        const int t = ti >> 2, i = ti & 3;
        const int orow = row0 + 16 * t + 4 * gs + i;
        if (orow < g.M) {
            float v = acc[t][i] + bias;
            float* dst = Cc + (size_t)orow * g.ldc + col;
            if (EPI == EPI_BIAS_SILU) v = silu_f(v);
            if (EPI == EPI_RESID_MASK) v = (*dst + v) * g.nmask[orow];
            *dst = v;
        }
    }
}
