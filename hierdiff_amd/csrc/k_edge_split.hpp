// Edge kernel for very small batches (every precision mode): ONE 32-edge tile per workgroup, its H output columns split over
// the four wavefronts.  Included through kernels.hpp after k_edge.hpp (same EdgeArgs, same tile tables, same weight image).
//
// Why: in k_edge a wavefront owns a whole 32-edge x H tile, i.e. a serial chain of H*H/64 fp32 MFMAs (1,024 at H = 256:
// 65.5 k matrix-pipe cycles = 29 us) plus prologue and epilogue - 43 us per launch however few tiles there are (18 - 25 us
// in the bf16 modes).  The
// reference's shipped sampling job is batch_size 2 (conf/sample/default.yaml:1-2): 56 tiles, 14 workgroups on a 256-CU
// chip.  Here every tile is spread over the four SIMDs of a CU: wavefront w computes columns [w*H/4, (w+1)*H/4) - a quarter
// of the MFMAs - with its W2 fragments going L2 -> registers directly (no LDS staging, no barrier in the loop: nothing is
// shared between the wavefronts there); the first-layer operand P = SiLU(A_i + B_j + r w_r + d0 w_d), which all four need
// in full, is built cooperatively - each wavefront a quarter of the K chunks - and shared through LDS (one barrier).
//
// Bit-identical to k_edge<H, COORD, PREC> by construction, so a molecule's bits still do not depend on the size of its batch:
//   * every output element sees the same MFMA chain (accumulator from b2, K chunks ascending; fp32: k-quad q, j;
//     bf16x3: k-step s, then head*head, tail*head, head*tail; bf16x6: h*L, h*M, m*M, h*H, m*H, l*H per 16-wide chunk)
//     on operands produced by the same expressions (k_edge's make_P / make_quad / make_quad_x6);
//   * the row dot with w_a / w_7 is one FMA chain per lane over the column tiles in ascending order - here it is handed
//     from wavefront to wavefront through LDS (three hand-offs) and finished by the same transposed reduction;
//   * gate / tanh head, masked per-node sums and the cross-half add are the same expressions on the same operands.
// tests/test_gpu_parity.py::test_small_batch_edge_kernel_is_bit_identical compares 1-, 2- and 6-molecule batches (this kernel)
// with the same molecules inside a 40-molecule batch (k_edge), in every precision mode.
#pragma once
#include "k_edge.hpp"

// LDS of the split form besides wrd_s: operand tile + hand-off scratch, carved from one byte buffer so that k_edge_mixed can
// place it in the dynamic LDS the whole-tile form uses for its W2 double buffer
template <int H, int PREC>
constexpr int edge_split_lds_bytes() {
    return (H / (PREC == 2 ? 16 : 32)) * (PREC == 2 ? 3 : 4) * 64 * 16 + (16 * 64 + 64 + 8 + 32 + 96 + 4 + 32) * 4;
}

// DEEP: the standalone launch (at most ~680 workgroups on 256 CUs: registers are free) keeps more K chunks of W2 fragments in
// flight than the body inside k_edge_mixed, which shares a 256-register budget with the whole-tile form
template <int H, bool COORD, int PREC, bool DEEP>
HD_DEVINL void edge_split_body(const EdgeArgs& a, char* lds, float* wrd_s, const int tile) {
    constexpr int KC = PREC == 2 ? 16 : 32;                          // K chunk width, as in k_edge
    constexpr int NCT = H / 32, NCW = NCT / 4, NCH = H / KC, CHF = PREC == 2 ? 24 * H : 32 * H, NQ = KC / 8;
    static_assert(NCT % 4 == 0, "column tiles are dealt to four wavefronts");
    constexpr int SL = PREC == 2 ? 3 : 4;                            // 16-byte slots per lane and chunk
    u32x4* opnd_s = reinterpret_cast<u32x4*>(lds);                   // [NCH * SL * 64] operand tile
    float* dot_x = reinterpret_cast<float*>(lds + NCH * SL * 64 * 16);   // [16 * 64] running row dots, handed from wavefront to wavefront
    float* att_s = dot_x + 16 * 64;                                  // [64] gate per (half, row slot) as wavefront 3 holds it
    uint32_t* seg_s = reinterpret_cast<uint32_t*>(att_s + 64);       // [8] segment byte of the 32 rows
    float* cs_phi = reinterpret_cast<float*>(seg_s + 8);             // [32] coordinate head scratch (wavefront 3)
    float* cs_tr = cs_phi + 32;                                      // [96]
    int* nan_p = reinterpret_cast<int*>(cs_tr + 96);
    float* rs_s = reinterpret_cast<float*>(nan_p + 4);               // [32] fp16x3: 1 / (row scale x W2 scale)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;

    for (int k = tid; k < 2 * H; k += 256) wrd_s[k] = a.wrd[k];
    for (int k = tid; k < H; k += 256) { wrd_s[2 * H + k] = a.b2[k]; wrd_s[3 * H + k] = a.wa[k]; }

    // per-row metadata (lanes n and n + 32 both describe row n), as in k_edge
    const int e = tile * 32 + n;
    const int ni = a.ei[e], nj = a.ej[e];
    const uint32_t segb = a.eseg[e];
    const f32x4 xi = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)ni * 4);
    const f32x4 xj = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)nj * 4);
    const f32x4 yi = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)ni * 4);
    const f32x4 yj = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)nj * 4);
    const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
    const float radial = dx * dx + dy * dy + dz * dz;
    const float ex = yi[0] - yj[0], ey = yi[1] - yj[1], ez = yi[2] - yj[2];
    const float d0 = ex * ex + ey * ey + ez * ez;
    const int pid_l = a.seg_part[tile * 32 + n];
    const int nseg = a.tile_nseg[tile];
    float f16_inv = 1.0f;                                            // fp16x3: this row's activation scale, as in k_edge
    if constexpr (PREC == 3) {
        const float nodes = a.abmax[2 * (size_t)ni] + a.abmax[2 * (size_t)nj + 1] + HD_F16_FLOOR;
        const float bound = __builtin_fmaf(radial, a.wrmax, __builtin_fmaf(d0, a.wdmax, nodes));
        const uint32_t eb = (__builtin_bit_cast(uint32_t, bound) >> 23) & 0xffu;
        f16_inv = __builtin_bit_cast(float, (eb - 13u) << 23);
        if (wave == 3 && hh == 0) rs_s[n] = f16_inv * a.w2s_inv;
    }
    if (wave == 3 && hh == 0) {
        reinterpret_cast<uint8_t*>(seg_s)[n] = (uint8_t)segb;
        if constexpr (COORD) {
            const float inv = ((segb != 255) ? 1.0f : 0.0f) / (sqrtf(radial + 1e-8f) + a.norm_constant);
            cs_tr[n * 3 + 0] = dx * inv; cs_tr[n * 3 + 1] = dy * inv; cs_tr[n * 3 + 2] = dz * inv;
        }
    }

    // first-layer operand of K chunk c for this lane's edge row (k = KC c + (KC/2) hh + 0 .. KC/2-1): rows requested two
    // chunks ahead, finished behind the MFMAs of the chunk before
    const float* Arow = a.AB + (size_t)ni * (2 * H) + (KC / 2) * hh;
    const float* Brow = a.AB + (size_t)nj * (2 * H) + H + (KC / 2) * hh;
    struct Raw { f32x4 a[NQ], b[NQ]; };
    auto load_raw = [&](int c, Raw& w) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            w.a[u] = *reinterpret_cast<const f32x4*>(Arow + KC * c + 4 * u);
            w.b[u] = *reinterpret_cast<const f32x4*>(Brow + KC * c + 4 * u);
        }
    };
    // operand registers: fp32 16 floats; bf16x3 head / tail of the two k-steps; bf16x6 head / middle / tail of the one k-step
    struct Opnd { float P[PREC == 0 ? 16 : 1]; u32x4 ph[2], pl[2], xh, xm, xl; };
    auto finish_P = [&](int c, const Raw& w, Opnd& o) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wrd_s + KC * c + (KC / 2) * hh + 4 * u);
            const f32x4 wd4 = *reinterpret_cast<const f32x4*>(wrd_s + H + KC * c + (KC / 2) * hh + 4 * u);
            if constexpr (PREC == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pre = w.a[u][j] + w.b[u][j];               // same operation order as k_edge's make_P
                    pre = __builtin_fmaf(radial, wr4[j], pre);
                    pre = __builtin_fmaf(d0, wd4[j], pre);
                    o.P[4 * u + j] = HD_F32_SILU(pre);
                }
            } else if constexpr (HD_TWOWAY(PREC)) {                  // k_edge's make_quad (scaled domain) + make_P_bf
                float pre[4], ev[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) pre[j] = w.a[u][j] + w.b[u][j];
#pragma unroll
                for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(radial, wr4[j], pre[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(d0, wd4[j], pre[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[j] = __builtin_amdgcn_exp2f(pre[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[j] = PREC == 3 ? __builtin_fmaf(ev[j], f16_inv, f16_inv) : 1.0f + ev[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[j] = __builtin_amdgcn_rcpf(ev[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) pre[j] *= ev[j];
                uint32_t hi[2], lo[2];
                split2<PREC == 3>(pre[0], pre[1], hi[0], lo[0]);
                split2<PREC == 3>(pre[2], pre[3], hi[1], lo[1]);
                o.ph[u >> 1][2 * (u & 1)] = hi[0]; o.ph[u >> 1][2 * (u & 1) + 1] = hi[1];
                o.pl[u >> 1][2 * (u & 1)] = lo[0]; o.pl[u >> 1][2 * (u & 1) + 1] = lo[1];
            } else {                                                 // k_edge's make_quad_x6
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pre = w.a[u][j] + w.b[u][j];
                    pre = __builtin_fmaf(radial, wr4[j], pre);
                    pre = __builtin_fmaf(d0, wd4[j], pre);
                    y[j] = HD_X6_SILU(pre);
                }
                uint32_t hi[2], mi[2], lo[2];
                bf16_split3(y[0], y[1], hi[0], mi[0], lo[0]);
                bf16_split3(y[2], y[3], hi[1], mi[1], lo[1]);
                o.xh[2 * u] = hi[0]; o.xh[2 * u + 1] = hi[1]; o.xm[2 * u] = mi[0]; o.xm[2 * u + 1] = mi[1];
                o.xl[2 * u] = lo[0]; o.xl[2 * u + 1] = lo[1];
            }
        }
    };
    // W2 fragments of this wavefront's column tiles, L2 -> registers, in k_edge's chunk-image layouts:
    //   fp32   [4 q][NCT][64 lanes][4 floats]                       NF = 4 NCW fragments per chunk: (q, k)
    //   bf16x3 [head|tail][2 k-steps][NCT][64 lanes][8 bf16]        NF = 4 NCW: (head / tail, s, k)
    //   bf16x6 [head|middle|tail][NCT][64 lanes][8 bf16]            NF = 3 NCW: (part, k)
    const int ct0 = wave * NCW;
    constexpr int NF = (PREC == 2 ? 3 : 4) * NCW;
    const char* wimg = reinterpret_cast<const char*>(a.W2img) + lane * 16;
    auto load_frags = [&](int c, u32x4 (&f)[NF]) {
        const char* base = wimg + (size_t)c * CHF * 4;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int x = i / NCW, k = i % NCW;
            unsigned off;
            if constexpr (PREC == 0) off = frag_off_f32(x * NCT + ct0 + k);                       // x = q
            else if constexpr (HD_TWOWAY(PREC)) off = (unsigned)(((((x >> 1) * 2 + (x & 1)) * NCT + ct0 + k) * 64) * 16);   // x = 2 hl + s
            else off = (unsigned)((x * NCT + ct0 + k) * 64 * 16);                                 // x = part
            f[i] = *reinterpret_cast<const u32x4*>(base + off);
        }
    };
    constexpr int RING0 = PREC == 0 ? 3 : (HD_TWOWAY(PREC) ? 4 : 6);       // chunks of fragments in flight ahead of the MFMAs
    constexpr int RING = DEEP ? (PREC == 0 ? 6 : (HD_TWOWAY(PREC) ? 6 : 8)) : RING0;
    u32x4 fr[RING][NF];
    static_for<0, (RING - 1 < NCH ? RING - 1 : NCH)>([&](auto Cc) { load_frags(decltype(Cc)::value, fr[decltype(Cc)::value]); });
    // The operand tile (32 edge rows x H) is the same for the four wavefronts: each builds a quarter of the K chunks
    // (c = wave, wave + 4, ...) and leaves the finished operand registers in LDS, [chunk][slot][lane] x 16 B (lane-linear:
    // conflict-free ds_write_b128 / ds_read_b128); lane l of every wavefront describes the same (row, half).
    constexpr int CPW = NCH / 4;                                     // chunks per wavefront
    {
        Raw raw[CPW];
#pragma unroll
        for (int i = 0; i < CPW; ++i) load_raw(4 * i + wave, raw[i]);
        __syncthreads();                                             // wrd_s, seg_s, cs_tr staged
#pragma unroll
        for (int i = 0; i < CPW; ++i) {
            const int c = 4 * i + wave;
            Opnd o;
            finish_P(c, raw[i], o);
            u32x4* dst = opnd_s + (size_t)c * SL * 64 + lane;
            if constexpr (PREC == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q * 64] = __builtin_bit_cast(u32x4, f32x4{o.P[4 * q], o.P[4 * q + 1], o.P[4 * q + 2], o.P[4 * q + 3]});
            } else if constexpr (HD_TWOWAY(PREC)) {
                dst[0] = o.ph[0]; dst[64] = o.ph[1]; dst[128] = o.pl[0]; dst[192] = o.pl[1];
            } else {
                dst[0] = o.xh; dst[64] = o.xm; dst[128] = o.xl;
            }
        }
    }
    __syncthreads();                                                 // operands of every chunk are in LDS
    auto read_op = [&](int c, u32x4 (&v)[SL]) {
#pragma unroll
        for (int q = 0; q < SL; ++q) v[q] = opnd_s[((size_t)c * SL + q) * 64 + lane];
    };
    u32x4 opv[2][SL];
    read_op(0, opv[0]);
    f32x16 acc[NCW];
#pragma unroll
    for (int k = 0; k < NCW; ++k) {
        const float b2v = PREC == 3 ? 0.0f : wrd_s[2 * H + 32 * (ct0 + k) + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = b2v;
    }
    static_for<0, NCH>([&](auto Cc) {
        constexpr int c = decltype(Cc)::value;
        if constexpr (c + RING - 1 < NCH) load_frags(c + RING - 1, fr[(c + RING - 1) % RING]);
        if constexpr (c + 1 < NCH) read_op(c + 1, opv[(c + 1) & 1]);
        u32x4(&f)[NF] = fr[c % RING];
        u32x4(&ov)[SL] = opv[c & 1];
        if constexpr (PREC == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < NCW; ++k)
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, ov[q])[j], __builtin_bit_cast(f32x4, f[q * NCW + k])[j], acc[k], 0, 0, 0);
        } else if constexpr (HD_TWOWAY(PREC)) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8 A_h = __builtin_bit_cast(bf16x8, ov[st]), A_l = __builtin_bit_cast(bf16x8, ov[2 + st]);
#pragma unroll
                for (int k = 0; k < NCW; ++k) {
                    const bf16x8 Wh = __builtin_bit_cast(bf16x8, f[(0 + st) * NCW + k]), Wl = __builtin_bit_cast(bf16x8, f[(2 + st) * NCW + k]);
                    acc[k] = mma16<PREC == 3>(A_h, Wh, acc[k]);
                    acc[k] = mma16<PREC == 3>(A_l, Wh, acc[k]);
                    acc[k] = mma16<PREC == 3>(A_h, Wl, acc[k]);
                }
            }
        } else {
            const bf16x8 A_h = __builtin_bit_cast(bf16x8, ov[0]), A_m = __builtin_bit_cast(bf16x8, ov[1]), A_l = __builtin_bit_cast(bf16x8, ov[2]);
#pragma unroll
            for (int k = 0; k < NCW; ++k) {
                const bf16x8 Wh = __builtin_bit_cast(bf16x8, f[0 * NCW + k]), Wm = __builtin_bit_cast(bf16x8, f[1 * NCW + k]),
                             Wt = __builtin_bit_cast(bf16x8, f[2 * NCW + k]);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, Wt, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, Wm, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, Wm, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, Wh, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, Wh, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_l, Wh, acc[k], 0, 0, 0);
            }
        }
    });

    // ---- epilogue.  acc[k][r] = row rho(r) = (r&3) + 8*(r>>2) + 4*hh, column 32*(ct0+k) + n.
    if constexpr (PREC == 3) {                                       // un-scale + bias, the same fma as k_edge
        f32x4 rsc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rsc[q] = *reinterpret_cast<const f32x4*>(rs_s + 8 * q + 4 * hh);
#pragma unroll
        for (int k = 0; k < NCW; ++k) {
            const float b2v = wrd_s[2 * H + 32 * (ct0 + k) + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = __builtin_fmaf(acc[k][r], rsc[r >> 2][r & 3], b2v);
        }
    }
#pragma unroll
    for (int k = 0; k < NCW; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (PREC == 0) acc[k][r] = HD_F32_SILU(acc[k][r]);
            else if constexpr (PREC == 2) acc[k][r] = HD_X6_SILU(acc[k][r]);
            else acc[k][r] = acc[k][r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[k][r]));   // scaled domain
        }
    // the row dot is one FMA chain per lane over ct = 0 .. NCT-1: wavefront w continues where w-1 stopped
    float dot[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dot[r] = 0.f;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            if (w > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dot[r] = dot_x[r * 64 + lane];
            }
#pragma unroll
            for (int k = 0; k < NCW; ++k) {
                const float wav = wrd_s[3 * H + 32 * (ct0 + k) + n];
#pragma unroll
                for (int r = 0; r < 16; ++r) dot[r] = __builtin_fmaf(acc[k][r], wav, dot[r]);
            }
            if (w < 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dot_x[r * 64 + lane] = dot[r];
            }
        }
        __syncthreads();
    }
    if (wave == 3) {
        // transpose-reduce over the 32 lanes of a half, exactly as in k_edge: lanes 2s, 2s+1 end with the dot of row slot s
        float rowdot;
        {
            float v8[8], v4[4], v2[2];
            const bool b4 = n & 16, b3 = n & 8, b2_ = n & 4, b1 = n & 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float send = b4 ? dot[k] : dot[k + 8];
                const float keep = b4 ? dot[k + 8] : dot[k];
                v8[k] = keep + __shfl_xor(send, 16);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float send = b3 ? v8[k] : v8[k + 4];
                const float keep = b3 ? v8[k + 4] : v8[k];
                v4[k] = keep + __shfl_xor(send, 8);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float send = b2_ ? v4[k] : v4[k + 2];
                const float keep = b2_ ? v4[k + 2] : v4[k];
                v2[k] = keep + __shfl_xor(send, 4);
            }
            {
                const float send = b1 ? v2[0] : v2[1];
                const float keep = b1 ? v2[1] : v2[0];
                rowdot = keep + __shfl_xor(send, 2);
            }
            rowdot += __shfl_xor(rowdot, 1);
        }
        const int my_slot = (n >> 1) & 15;
        if constexpr (!COORD) {
            float att_mine = 1.0f;
            if (a.attention) {
                const float ba = a.ba_ptr ? *a.ba_ptr : a.ba;
                if constexpr (!HD_TWOWAY(PREC)) att_mine = sigmoid_f(rowdot + ba);
                else att_mine = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(rowdot + ba));   // scaled domain
            }
            att_s[lane] = att_mine;
            const bool tile_has_nan = __builtin_amdgcn_ballot_w64(rowdot != rowdot) != 0;
            if (lane == 0) *nan_p = tile_has_nan ? 1 : 0;
        } else {
            // phi of row rho(slot) is in lanes 2 slot, 2 slot + 1 of half hh; then lane n handles row n (k_edge's coordinate head)
            if ((n & 1) == 0) cs_phi[(my_slot & 3) + 8 * (my_slot >> 2) + 4 * hh] = rowdot;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (hh == 0) {
                const float phi = cs_phi[n];
                const float sc = a.use_tanh ? tanhf(phi) * a.coords_range : phi;
                cs_tr[n * 3 + 0] *= sc;
                cs_tr[n * 3 + 1] *= sc;
                cs_tr[n * 3 + 2] *= sc;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < nseg) {
                const uint8_t* sb = reinterpret_cast<const uint8_t*>(seg_s);
                float sx = 0.f, sy = 0.f, sz = 0.f;
                for (int rr = 0; rr < 32; ++rr) {
                    if (sb[rr] == lane) { sx += cs_tr[rr * 3]; sy += cs_tr[rr * 3 + 1]; sz += cs_tr[rr * 3 + 2]; }
                }
                const f32x4 o = {sx, sy, sz, 0.f};
                *reinterpret_cast<f32x4*>(a.part + (size_t)pid_l * 4) = o;   // lane < nseg <= 32: pid_l is segment `lane`'s id
            }
        }
    }
    if constexpr (COORD) return;
    __syncthreads();                                                 // att_s, nan_s published
    {
        uint32_t sw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sw[q] = seg_s[2 * q + hh];
        float w[16];
        int sg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sg[r] = (sw[r >> 2] >> (8 * (r & 3))) & 255;
            const float att = att_s[(lane & 32) | (2 * r)];
            w[r] = (sg[r] != 255) ? att : 0.0f;
        }
        const bool tile_has_nan = *nan_p != 0;
        for (int s = 0; s < nseg; ++s) {
            float ws[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ws[r] = (sg[r] == s) ? w[r] : 0.0f;
            float* dst = a.part + (size_t)__builtin_amdgcn_readlane(pid_l, s) * H + 32 * ct0 + n;
            float sums[NCW];
            if (__builtin_expect(tile_has_nan, 0)) {
#pragma unroll
                for (int k = 0; k < NCW; ++k) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum = (sg[r] == s) ? __builtin_fmaf(ws[r], acc[k][r], sum) : sum;
                    sums[k] = sum;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NCW; ++k) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum = __builtin_fmaf(ws[r], acc[k][r], sum);
                    sums[k] = sum;
                }
            }
#pragma unroll
            for (int k = 0; k < NCW; ++k) sums[k] = xhalf_sum(sums[k]);
            if (hh == 0) {
#pragma unroll
                for (int k = 0; k < NCW; ++k) dst[32 * k] = sums[k];
            }
        }
    }
}

template <int H, bool COORD, int PREC>
__global__ __launch_bounds__(256) void k_edge_split(EdgeArgs a) {
    __shared__ __attribute__((aligned(16))) float wrd_s[4 * H];     // [w_r | w_d | b2 | wa]
    __shared__ __attribute__((aligned(16))) char lds[edge_split_lds_bytes<H, PREC>()];
    edge_split_body<H, COORD, PREC, true>(a, lds, wrd_s, blockIdx.x);      // grid = n_tiles
}

// Whole-tile and column-split workgroups in ONE launch, for topologies between one and three tiles per SIMD.  There a
// k_edge launch leaves the chip unevenly loaded: all its workgroups are resident at once (two per CU at most), a CU that got
// two takes twice as long as a CU that got one, and nothing is left to back-fill (B = 64 at N = 30: 436 workgroups on 512
// slots, 81 us against 43 us for one workgroup per CU).  Here the first a.n_wg blocks are whole-tile workgroups (four tiles
// each, k_edge's body) - a multiple of the CU count, so every CU gets the same number - and the remaining tiles follow as
// column-split single-tile workgroups (k_edge_split's body), which the dispatcher deals out as slots free up.  Both bodies are
// bit-identical per tile, so which one computes a tile does not show in the result.  Dynamic LDS = the larger of the two needs.
template <int H, bool COORD, int PREC>
__global__ __launch_bounds__(256, 2) void k_edge_mixed(EdgeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ __attribute__((aligned(16))) float wrd_s[4 * H];
    const int bid = blockIdx.x;
    if (bid < a.n_wg) edge_tile_body<H, COORD, PREC, 0>(a, smem, wrd_s, bid, a.n_wg);
    else edge_split_body<H, COORD, PREC, false>(a, reinterpret_cast<char*>(smem), wrd_s, 4 * a.n_wg + (bid - a.n_wg));
}
