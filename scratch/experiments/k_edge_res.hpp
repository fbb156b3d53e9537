// Exact-fp32 edge kernel with the second-layer weights RESIDENT IN REGISTERS (round 4): one persistent workgroup per CU walks a
// contiguous range of 32-edge tiles, every tile's H output columns split over the H/32 wavefronts of the workgroup.  Included
// through kernels.hpp after k_edge.hpp (same EdgeArgs, same tile tables, same fp32 weight image).
//
// Why.  k_edge gives a wavefront a whole 32-edge x H tile: 1,024 fp32 MFMAs at H = 256, all of a launch's tiles resident at once,
// two per SIMD.  A launch is therefore as slow as its fullest SIMD, in whole tiles: 1.7 tiles per SIMD (B = 64 at N = 30, and
// every GEOM-sized batch of 256) cost 2.0 tile times - 15 % of the chip idle - and 1.14 (config 5) cost 2.0 as well.  Splitting
// a tile's COLUMNS over wavefronts removes the quantisation (the unit of work per SIMD becomes a fraction of a tile), but
// k_edge_split pays for it with W2 traffic: each wavefront pulls its column slice of W2 (H*H/4 floats) from L2 for every tile.
// Here the slice stays in registers for the whole launch: wavefront w of NW owns H/NW columns, i.e. H*H/NW floats of W2 =
// H*H/(64 NW) registers per lane (H = 256, NW = 4: 256 registers - the workgroup runs one wavefront per SIMD with the 512-entry
// register file to itself; NW = 8: 128 of the 256 registers of two wavefronts per SIMD).  Per tile a wavefront then runs
// H*H/(64 NW) MFMAs whose B operand is a register it already holds and whose A operand - the first-layer activations P, the
// same for all wavefronts - is read from LDS, where the wavefronts built it together (each its share of the K chunks).  No
// W2 stream, no per-chunk barrier: ONE barrier per tile.
// Tiles are dealt to the launch's workgroups as contiguous ranges whose lengths differ by at most one, so a launch takes
// ceil(tiles / CUs) tile times whatever the batch size.
//
// Arithmetic.  Every element of the second layer sees k_edge's MFMA chain (accumulator from b2, K chunks ascending, k-quad q, j)
// on operands from k_edge's make_P, SiLU as there: M[e][c] has k_edge's bits.  The row dot with w_a / w_7 is reduced
// differently: each wavefront reduces its own 32 columns (k_edge's transposed reduction), the H/32 partial dots meet in LDS and
// every wavefront adds them in wavefront order.  That order is a function of the tile alone - not of the batch, the launch
// geometry or the CU - so a molecule's bits still do not depend on its batch neighbours, the rank or the world size; they DO
// differ (in the last bits of the gate) from k_edge's, which is why the fp32 mode uses this kernel for every launch of the
// widths it supports (H = 128, 256) and never mixes the two forms.
#pragma once
#include "k_edge.hpp"

// measurement build: per-wave cycle stamps of one steady-state tile (the third of a workgroup) in k_node.hpp's trace buffer
#ifdef HD_DEBUG_KERNELS
#define HD_RSTAMP(k)                                                                                                    \
    do {                                                                                                                \
        if (!COORD && t == t0 + 2) {                                                                                    \
            const long long ts_ = __builtin_readcyclecounter();                                                         \
            if (lane == 0 && blockIdx.x < 512 && wave < 8) hd_ntrace[((size_t)blockIdx.x * 8 + wave) * HD_NTRACE_STAMPS + (k)] = ts_; \
        }                                                                                                               \
    } while (0)
#else
#define HD_RSTAMP(k) do { } while (0)
#endif

// acc += a x b (v_mfma_f32_32x32x2_f32) with the B operand in an AGPR.  With 256 resident W2 registers per lane (H = 256,
// NW = 4) the slice has to live in the accumulator half of the 512-entry register file; hipcc would keep it there but copy
// every value to a VGPR in front of its MFMA (one v_accvgpr_read per MFMA) and park the accumulators in AGPRs instead, so the
// instruction is written out with an "a" operand.  Inline asm is invisible to the hazard recogniser: mfma_drain() below
// covers the MFMA-write -> VALU-read distance once per tile; dependent MFMAs on the same accumulator interlock in hardware.
HD_DEVINL void mfma_f32_b_agpr(f32x16& acc, float a, float b) {
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
HD_DEVINL void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 3" ::: "memory"); }

// LDS: operand tile, double-buffered: [2][H/32 chunks][4 q][64 lanes] x 16 B; partial row dots [2][32 rows][NW]; per wavefront:
// masked gates of every segment [32][32] floats (the coordinate head's scaled component of every row lives in its first row)
template <int H, int NW>
constexpr int edge_res_lds_bytes() { return 2 * (H / 32) * 4 * 64 * 16 + 2 * 32 * NW * 4 + NW * 32 * 32 * 4; }

// GLOAD = false: radial, the initial distance and the unit direction of every edge row are computed from the coordinate
// gathers (k_edge's expressions) and stored in the topology's `geom` / `d0tab` tables; GLOAD = true: they are read back -
// the coordinates change once per block (k_xupd), so of a block's three edge launches only the first gathers them.
template <int H, bool COORD, int NW, bool GLOAD>
__global__ __launch_bounds__(64 * NW, NW <= 4 ? 1 : 2) void k_edge_res(EdgeArgs a) {
    constexpr int NCH = H / 32, NCT = H / 32;
    constexpr int CW = NCT / NW;           // 32-column tiles per wavefront
    constexpr int CB = NCH / NW;           // K chunks of the operand each wavefront builds
    constexpr int NT = 64 * NW;
    static_assert(CW >= 1 && CW * NW == NCT, "NW must divide H/32");
    constexpr bool WAGPR = H * H / NW / 64 > 128;    // more than 128 W2 registers per lane: they live in AGPRs (mfma_f32_b_agpr)
    constexpr int OPB = NCH * 4 * 64;      // f32x4 slots per operand buffer
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    __shared__ __attribute__((aligned(16))) float wrd_s[4 * H];       // [w_r | w_d | b2 | wa]
    f32x4* opnd = reinterpret_cast<f32x4*>(smem_r);
    float* rowpart = reinterpret_cast<float*>(smem_r + 2 * OPB * 16);  // [2][32][NW]
    float* gs_all = rowpart + 2 * 32 * NW;                              // [NW][32 segments][32 rows]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    float* gs = gs_all + wave * 32 * 32;                               // wave-private

    // this workgroup's tile range: XCD-aware (block b runs on XCD b % 8; the workgroups of an XCD take neighbouring ranges =
    // the same molecules' AB rows in that XCD's L2), lengths differ by at most one
    int t0, t1;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        if (slot >= len) return;
        const long long L = start + slot;
        t0 = (int)(L * a.n_tiles / nblk);
        t1 = (int)((L + 1) * a.n_tiles / nblk);
        if (t0 >= t1) return;
    }

    // per row slot n of a tile (hd_topology_create): {receiving node, sending node, part id of SEGMENT n, segment byte of ROW n |
    // number of segments << 8}
    struct Meta { int ni, nj, pid; uint32_t sn; };
    auto load_meta = [&](int tile) {
        const i32x4 v = a.emeta[(unsigned)(tile * 32 + n)];
        Meta m;
        m.ni = v[0]; m.nj = v[1]; m.pid = v[2]; m.sn = (uint32_t)v[3];
        return m;
    };
    struct Gath { f32x3 xi, xj, yi, yj; f32x4 gm; float d0; f32x4 pa[CB][4], pb[CB][4]; };
    auto issue_gathers = [&](const Meta& m, int tile, Gath& g) {
        if constexpr (GLOAD) {
            g.gm = *reinterpret_cast<const f32x4*>(a.geom + (size_t)(unsigned)(tile * 32 + n) * 4);
            g.d0 = a.d0tab[(unsigned)(tile * 32 + n)];
        } else {
            g.xi = *reinterpret_cast<const f32x3*>(a.xcur + (size_t)(unsigned)m.ni * 4);
            g.xj = *reinterpret_cast<const f32x3*>(a.xcur + (size_t)(unsigned)m.nj * 4);
            g.yi = *reinterpret_cast<const f32x3*>(a.x0 + (size_t)(unsigned)m.ni * 4);
            g.yj = *reinterpret_cast<const f32x3*>(a.x0 + (size_t)(unsigned)m.nj * 4);
        }
        const float* Arow = a.AB + (size_t)(unsigned)m.ni * (2 * H) + 32 * CB * wave + 16 * hh;   // this wavefront builds K chunks CB wave ..
        const float* Brow = a.AB + (size_t)(unsigned)m.nj * (2 * H) + H + 32 * CB * wave + 16 * hh;
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g.pa[i][u] = *reinterpret_cast<const f32x4*>(Arow + 32 * i + 4 * u);
                g.pb[i][u] = *reinterpret_cast<const f32x4*>(Brow + 32 * i + 4 * u);
            }
    };
    // first-layer activations of this wavefront's K chunks for this lane's edge row (k_edge's make_P), left in LDS for every
    // wavefront; the unit direction of the edge for the coordinate head (egnn_new.py:96-99; zero for padding rows)
    auto build_operand = [&](const Meta& m, const Gath& g, int tile, int buf, float (&dir)[3]) {
        float radial, d0;
        if constexpr (GLOAD) {
            radial = g.gm[3]; d0 = g.d0;
            dir[0] = g.gm[0]; dir[1] = g.gm[1]; dir[2] = g.gm[2];
        } else {
            const float dx = g.xi[0] - g.xj[0], dy = g.xi[1] - g.xj[1], dz = g.xi[2] - g.xj[2];
            radial = dx * dx + dy * dy + dz * dz;
            const float ex = g.yi[0] - g.yj[0], ey = g.yi[1] - g.yj[1], ez = g.yi[2] - g.yj[2];
            d0 = ex * ex + ey * ey + ez * ez;
            const float inv = (((m.sn & 255) != 255) ? 1.0f : 0.0f) / (sqrtf(radial + 1e-8f) + a.norm_constant);
            dir[0] = dx * inv; dir[1] = dy * inv; dir[2] = dz * inv;
            if (wave == 0 && hh == 0) {
                const f32x4 o = {dir[0], dir[1], dir[2], radial};
                *reinterpret_cast<f32x4*>(a.geom + (size_t)(unsigned)(tile * 32 + n) * 4) = o;
                a.d0tab[(unsigned)(tile * 32 + n)] = d0;
            }
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = CB * wave + i;
            f32x4* dst = opnd + buf * OPB + c * 4 * 64 + lane;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wrd_s + 32 * c + 16 * hh + 4 * u);
                const f32x4 wd4 = *reinterpret_cast<const f32x4*>(wrd_s + H + 32 * c + 16 * hh + 4 * u);
                f32x4 P;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float pre = g.pa[i][u][j] + g.pb[i][u][j];
                    pre = __builtin_fmaf(radial, wr4[j], pre);
                    pre = __builtin_fmaf(d0, wd4[j], pre);
                    P[j] = HD_F32_SILU(pre);
                }
                dst[u * 64] = P;
            }
        }
    };

    // requests in the order their data is needed (loads return in issue order): the first tile's table and rows, the staged
    // vectors, and only then the wavefront's column slice of W2 (first used by the first MFMA)
    Meta mC = load_meta(t0);
    Meta mN = load_meta(t0 + 1 < t1 ? t0 + 1 : t1 - 1);
    float dirC[3] = {0.f, 0.f, 0.f}, dirN[3] = {0.f, 0.f, 0.f};
    Gath g0;
    issue_gathers(mC, t0, g0);
    for (int k = tid; k < 2 * H; k += NT) wrd_s[k] = a.wrd[k];
    for (int k = tid; k < H; k += NT) { wrd_s[2 * H + k] = a.b2[k]; wrd_s[3 * H + k] = a.wa[k]; }
    // this wavefront's column slice of W2, fragment order of the fp32 chunk image [c][4 q][NCT][64 lanes][4 j]: stays in
    // registers for the whole launch
    f32x4 Wr[NCH][4][CW];
    {
        const float* wimg = a.W2img + (size_t)(CW * wave) * 256 + lane * 4;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < CW; ++k)
                    Wr[c][q][k] = *reinterpret_cast<const f32x4*>(wimg + (size_t)c * 32 * H + (size_t)(q * NCT + k) * 256);
    }
    __syncthreads();                       // wrd_s staged
    build_operand(mC, g0, t0, 0, dirC);
    __syncthreads();                       // operand of the first tile complete
    float b2v[CW], wav[CW];
#pragma unroll
    for (int k = 0; k < CW; ++k) { b2v[k] = wrd_s[2 * H + 32 * (CW * wave + k) + n]; wav[k] = wrd_s[3 * H + 32 * (CW * wave + k) + n]; }

#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        HD_RSTAMP(0);
        // requests for the next tile (consumed behind this tile's MFMAs) and the tile table of the one after
        const int tn = t + 1 < t1 ? t + 1 : t1 - 1;
        Gath gn;
        issue_gathers(mN, tn, gn);

        // ---- second layer: acc[k][32 x 32] += P[32 x H] W2[:, this wavefront's columns]^T, A fragments from LDS a unit ahead
        f32x16 acc[CW];
#pragma unroll
        for (int k = 0; k < CW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = b2v[k];
        {
            const unsigned ob = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(opnd + buf * OPB) + lane * 16;
            // one fragment (4 CW MFMAs = 256 CW matrix-pipe cycles) ahead: the next fragment is requested before the current one's
            // MFMAs issue; the sched_barrier pins that order (left alone hipcc moves the MFMAs in front of the request, reuses
            // the fragment registers and waits out the LDS latency once per fragment)
            f32x4 f0, f1;
            asm volatile("ds_read_b128 %0, %1" : "=&v"(f0) : "v"(ob));
            HD_RSTAMP(1);
            static_for<0, 4 * NCH>([&](auto Uc) {
                constexpr int u = decltype(Uc)::value, c = u >> 2, q = u & 3;
                f32x4& cur = (u & 1) ? f1 : f0;
                f32x4& nxt = (u & 1) ? f0 : f1;
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur));
                if constexpr (u + 1 < 4 * NCH) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(nxt) : "v"(ob), "i"((u + 1) * 1024));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < CW; ++k) {
                        if constexpr (WAGPR) mfma_f32_b_agpr(acc[k], cur[j], Wr[c][q][k][j]);
                        else acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[j], Wr[c][q][k][j], acc[k], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (WAGPR) mfma_drain();
        }

        HD_RSTAMP(2);
        const Meta mNN = load_meta(t + 2 < t1 ? t + 2 : t1 - 1);       // lands behind the epilogue; used at the top of the next tile
        // ---- epilogue, first half.  acc[k][r] = row rho(r) = (r&3) + 8*(r>>2) + 4*hh, column 32*(CW wave + k) + n.
        float dot[16];
#pragma unroll
        for (int k = 0; k < CW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float mv = HD_F32_SILU(acc[k][r]);
                acc[k][r] = mv;
                dot[r] = k == 0 ? mv * wav[0] : __builtin_fmaf(mv, wav[k], dot[r]);
            }
        float rowdot;                      // this wavefront's columns of the row dot: k_edge's transposed reduction
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
        const int my_slot = (n >> 1) & 15;                              // lanes 2s, 2s+1 of half hh hold row rho(s) + 4 hh
        const int my_row = (my_slot & 3) + 8 * (my_slot >> 2) + 4 * hh;
        if ((n & 1) == 0) rowpart[(buf * 32 + my_row) * NW + wave] = rowdot;
        HD_RSTAMP(3);

        // ---- operand of the next tile into the other buffer (its last readers passed the previous barrier)
        if (t + 1 < t1) build_operand(mN, gn, tn, buf ^ 1, dirN);
        HD_RSTAMP(4);
        __syncthreads();                   // partial row dots of this tile + operand of the next one

        // ---- epilogue, second half.  Lane n completes the dot of row n (partials added in wavefront order), turns it into the
        // row's gate (zero for padding rows) / coordinate scale, and the per-node sums run over this wavefront's columns.
        float tot;
        {
            const float* rpn = rowpart + (buf * 32 + n) * NW;
            if constexpr (NW % 4 == 0) {
                tot = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < NW; w4 += 4) {
                    const f32x4 p4 = *reinterpret_cast<const f32x4*>(rpn + w4);
                    tot = w4 == 0 ? p4[0] : tot + p4[0];
                    tot += p4[1]; tot += p4[2]; tot += p4[3];
                }
            } else {
                tot = rpn[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) tot += rpn[w];
            }
        }
        HD_RSTAMP(5);
        const uint32_t segb = mC.sn & 255;
        const int nseg = __builtin_amdgcn_readfirstlane((int)(mC.sn >> 8));
        if constexpr (!COORD) {
            float gate = 1.0f;
            if (a.attention) {
                const float ba = a.ba_ptr ? *a.ba_ptr : a.ba;
                gate = sigmoid_f(tot + ba);
            }
            gate = (segb != 255) ? gate : 0.0f;
            // a tile holding a NaN row dot takes select-based sums (k_edge: a poisoned molecule must not leak into its tile
            // neighbours through 0 * NaN)
            const bool tile_has_nan = __builtin_amdgcn_ballot_w64(tot != tot) != 0;
            // masked gates of every segment, one LDS row of 32 per segment: lane (hh, n) then reads the weights of its 16
            // accumulator rows 8q + 4hh + k as four 16-byte words per segment (same address over a half: broadcast)
            if (hh == 0) {
                for (int s = 0; s < nseg; ++s) gs[s * 32 + n] = (segb == (uint32_t)s) ? gate : 0.0f;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int s = 0; s < nseg; ++s) {
                float sum[CW];
#pragma unroll
                for (int k = 0; k < CW; ++k) sum[k] = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(gs + s * 32 + 8 * q + 4 * hh);
                    if (__builtin_expect(tile_has_nan, 0)) {
                        // rows outside the segment (weight exactly 0) are skipped instead of multiplied: 0 * NaN would leak
                        // into this node's sum; a NaN weight (the poisoned molecule's own rows) is != 0 and stays in
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int k = 0; k < CW; ++k) sum[k] = (g4[j] != 0.0f) ? __builtin_fmaf(g4[j], acc[k][4 * q + j], sum[k]) : sum[k];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int k = 0; k < CW; ++k) sum[k] = __builtin_fmaf(g4[j], acc[k][4 * q + j], sum[k]);
                    }
                }
                float* dst = a.part + (size_t)__builtin_amdgcn_readlane(mC.pid, s) * H + 32 * CW * wave + n;
#pragma unroll
                for (int k = 0; k < CW; ++k) {
                    const float v = xhalf_sum(sum[k]);
                    if (hh == 0) dst[32 * k] = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else {
            // coordinate head: wavefront c < 3 owns component c.  Lane n scales its row's unit direction by tanh(phi) * range
            // (egnn_new.py:100-104), the 32 values meet in the wave-private LDS row, and lane s adds the rows of segment s in
            // ascending row order - a sum that depends on the segment's rows alone, not on where the piece sits in the tile
            if (wave < 3) {
                const float sc = a.use_tanh ? tanhf(tot) * a.coords_range : tot;
                const float mine = (wave == 0 ? dirC[0] : (wave == 1 ? dirC[1] : dirC[2])) * sc;
                uint32_t* gsu = reinterpret_cast<uint32_t*>(gs + 32);              // segment byte of every row, as integers
                if (hh == 0) { gs[n] = mine; gsu[n] = segb; }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(gs + 4 * q);
                    const u32x4 s4 = *reinterpret_cast<const u32x4*>(gsu + 4 * q);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float add = sum + v4[k];
                        sum = (s4[k] == (uint32_t)n) ? add : sum;
                    }
                }
                if (lane < nseg) a.part[(size_t)mC.pid * 4 + wave] = sum;          // lane < nseg <= 32: pid is segment `lane`'s id
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        HD_RSTAMP(6);
        mC = mN; mN = mNN;
#pragma unroll
        for (int k = 0; k < 3; ++k) dirC[k] = dirN[k];
    }
}
