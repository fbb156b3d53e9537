// EXPERIMENT (round 2), NOT part of the library: kept for the record, with its measurements (profiles/history/r02_x6p_experiment.log).
//   Result: bit-identical to k_edge<256, *, 2> (GPU parity / shard-equality tests green) and NOT faster: 173 us (GCL) / 164 us
//   (COORD) with two W2 buffers, 178 / 171 us with three, against 153-165 us for k_edge<256, *, 2>.  Ablations of the
//   three-buffer version: without the riding VALU work 126 us; without the W2 stream (barrier kept) 145 us; without the barrier
//   (stream kept) 169 us; neither stream nor riding work 112 us.  I.e. the LDS-DMA stream of W2 - 24 KB per chunk, every CU the
//   same lines at the same time in this lock-step persistent form - costs 19 %, the riding work 25 %; a 5-instruction SiLU
//   changes 3 us, the MFMA : VALU interleave ratio (2 / 4 / 6 / 8) nothing.  With one wavefront per SIMD there is no second
//   wavefront to cover the barrier and the stream waits, and that costs more than the in-wave co-execution gains.
//   To build it again: include it from kernels.hpp, add `int dump_part` to EdgeArgs (= n_parts, part buffers one row larger)
//   and launch grid = min(n_wg, CUs) with (3 * 24 * 256 + 4 * 2048) * 4 bytes of dynamic LDS.
//
// Persistent, software-pipelined form of the bf16x6 edge kernel (H = 256).  Included through kernels.hpp.
//
// Same arithmetic and the same bits as k_edge<256, COORD, 2> (same tile tables, same operation order per value), arranged so
// that the VALU work of a tile co-executes with MFMAs of the SAME wavefront: on gfx950 VALU instructions co-execute with the
// bf16 MFMAs of their own instruction stream (about four per MFMA are free, scratch/mb/coissue.hip), hardly with those of the
// other wavefront of a SIMD (SQ_VALU_MFMA_COEXEC_CYCLES: 0.38 of the MFMA-busy cycles in k_edge, whose prologue / epilogue -
// 43 % of its VALU work - has no MFMAs of its own to ride under).  One wavefront per SIMD (512 registers), persistent
// workgroups walking a contiguous run of tiles, two accumulator sets:
//   * the epilogue of tile t-1 (SiLU + attention / coordinate dot, row reduction, gate, per-node sums, stores) is cut into
//     slices that ride under the 128 six-MFMA stages of tile t;
//   * the next tile's tables, coordinates and first AB rows are requested in chunks 11-15, so a tile has no prologue;
//   * operand generation of the next K chunk (SiLU, three-way bf16 split) one value per stage.
// (The same design for the exact-fp32 kernel, scratch/experiments/k_edge_f32p.hpp, is slower than k_edge: nothing co-executes
// with the fp32 MFMA.  For bf16x3 there are 12 VALU instructions per MFMA - VALU-bound either way.)
// LDS reads are inline asm throughout the loop and requested a stage or two ahead (fragments: two stages; everything the
// riding work needs - gathered AB rows, w_r / w_d, wa - one stage), released by one counted lgkmcnt wait per stage.
// AB rows travel global -> LDS by DMA into a per-wave ring (an asm load with a VGPR destination is not safe here) and are
// requested 2-3 chunks before use.  VMEM bookkeeping per chunk, in issue order: [next-tile loads in chunks 11 / 12] + 6 stream
// pieces of chunk c+2 (three W2 buffers: with one wavefront per SIMD a chunk lasts ~1 us, less than a loaded L2 round trip) +
// rows of chunk c+3 (quad 0 behind stage 1, quad 1 behind stage 5): the chunk barrier waits "at most 14 outstanding", the
// quad-0 read in stage 7 "at most 12".
#pragma once
#include "k_edge.hpp"

template <unsigned O0, unsigned O1>
HD_DEVINL void x6p_read2(f32x4& x, f32x4& y, unsigned addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(x), "=&v"(y) : "v"(addr), "i"(O0), "i"(O1));
}
template <int N, unsigned O0, unsigned O1>
HD_DEVINL void x6p_read2_after_vm(f32x4& x, f32x4& y, unsigned addr) {
    asm volatile("s_waitcnt vmcnt(%3)\n\tds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %2 offset:%5"
                 : "=&v"(x), "=&v"(y) : "v"(addr), "i"(N), "i"(O0), "i"(O1));
}
template <unsigned O>
HD_DEVINL void x6p_read1(float& x, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(x) : "v"(addr), "i"(O)); }
struct X6Aux { f32x4 qa, qb, wr, wd; float wavA, wavB; };
template <int N>
HD_DEVINL void x6p_wait4(bf16x8 (&f)[4], X6Aux& x) {
    asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(x.qa), "+v"(x.qb), "+v"(x.wr), "+v"(x.wd),
                 "+v"(x.wavA), "+v"(x.wavB) : "i"(N));
}
template <int N>
HD_DEVINL void x6p_wait2(bf16x8 (&f)[2], X6Aux& x) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(f[0]), "+v"(f[1]), "+v"(x.qa), "+v"(x.qb), "+v"(x.wr), "+v"(x.wd), "+v"(x.wavA), "+v"(x.wavB) : "i"(N));
}

#ifndef X6P_ABL
#define X6P_ABL 0           // measurement builds only: 1 = no riding work, 2 = no barrier / W2 stream, 4 = no forced interleave
#endif
#ifndef X6P_VPM
#define X6P_VPM 4
#endif
template <bool COORD, int VPM = X6P_VPM>
__global__ __launch_bounds__(256, 1) void k_edge_x6p(EdgeArgs a) {
    constexpr int H = 256, NCT = 8, NP = 4, NCH = 16, CHF = 24 * H, GLW = CHF / (4 * 256);      // 16-wide K chunks, 6 pieces per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wbuf = smem;                                           // [3][CHF]: the W2 stream runs two chunks ahead
    __shared__ __attribute__((aligned(16))) float wrd_s[4 * H];   // [w_r | w_d | b2 | wa]
    float* rows_all = smem + 3 * CHF;                             // per wave: [2 chunk parities][4 slots] x 1 KiB (asm access only)
    __shared__ __attribute__((aligned(16))) float scr_s[4 * 2 * 144];   // per wave, per tile parity: 32 phi + 96 trans + 8 seg words

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    float* scr_w = scr_s + wave * 2 * 144;
    const float att_bias = a.ba_ptr ? *a.ba_ptr : a.ba;
    const unsigned rows_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(rows_all + wave * 2048);
    const unsigned rows_lane = rows_lds + lane * 16;
    const unsigned wrd_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)wrd_s;
    const unsigned wrd_k = wrd_lds + 32 * hh, wrd_n = wrd_lds + 4 * n;      // this lane's k-slice of w_r / w_d; its column of b2 / wa

    int wt_lo, wt_cnt;                                            // contiguous share of the workgroup-tiles
    {
        const int G = gridDim.x, b = blockIdx.x, q = a.n_wg / G, r = a.n_wg % G;
        wt_lo = b * q + (b < r ? b : r);
        wt_cnt = q + (b < r ? 1 : 0);
    }
    if (wt_cnt == 0) return;
    const int wt_last = wt_lo + wt_cnt - 1;

    for (int k = tid; k < 2 * H; k += 256) wrd_s[k] = a.wrd[k];
    for (int k = tid; k < H; k += 256) { wrd_s[2 * H + k] = a.b2[k]; wrd_s[3 * H + k] = a.wa[k]; }
    auto issue_chunk = [&](int c, int buf) {
        const float* src = a.W2img + (size_t)c * CHF + wave * (GLW * 256) + lane * 4;
        float* dst = wbuf + buf * CHF + wave * (GLW * 256);
        static_for<0, GLW>([&](auto U) {
            constexpr int u = decltype(U)::value;
            glds16o<(u & 3) * 1024>(src + (u >> 2) * 1024, dst + (u >> 2) * 1024);
        });
    };

    struct Tile {
        const float* Arow;
        const float* Brow;
        int ni, nj, nseg, pid;
        uint32_t segb;
        float radial, d0;
    };
    auto tile_meta = [&](Tile& t, int wt) {
        const int tile = wt * 4 + wave;                          // the tables cover 4 n_wg tiles (padding tiles: no segments)
        const int e = tile * 32 + n;
        t.ni = a.ei[e]; t.nj = a.ej[e]; t.segb = a.eseg[e];
        t.pid = a.seg_part[e];
        t.nseg = a.tile_nseg[tile];
        t.Arow = a.AB + (size_t)t.ni * (2 * H) + 8 * hh;
        t.Brow = a.AB + (size_t)t.nj * (2 * H) + H + 8 * hh;
    };
    auto tile_geom = [&](Tile& t, f32x4 xi, f32x4 xj, f32x4 yi, f32x4 yj, int par) {
        const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
        t.radial = dx * dx + dy * dy + dz * dz;
        const float ex = yi[0] - yj[0], ey = yi[1] - yj[1], ez = yi[2] - yj[2];
        t.d0 = ex * ex + ey * ey + ez * ez;
        float* sc = scr_w + par * 144;                           // lanes n and n + 32 describe the same row: same stores
        reinterpret_cast<uint8_t*>(sc + 128)[n] = (uint8_t)t.segb;
        if constexpr (COORD) {
            const float inv = ((t.segb != 255) ? 1.0f : 0.0f) / (sqrtf(t.radial + 1e-8f) + a.norm_constant);
            sc[32 + n * 3 + 0] = dx * inv; sc[32 + n * 3 + 1] = dy * inv; sc[32 + n * 3 + 2] = dz * inv;
        }
    };
    // row gathers of K chunk c of tile t, quad u: A quad -> slot 2u, B quad -> slot 2u+1 of ring parity c & 1
    auto rows_issue = [&](const Tile& t, int c, int u) {
        vm_glds2(t.Arow + 16 * c + 4 * u, t.Brow + 16 * c + 4 * u, rows_lds + (c & 1) * 4096 + (2 * u) * 1024,
                 rows_lds + (c & 1) * 4096 + (2 * u + 1) * 1024);
    };

    // ---- epilogue pieces: the arithmetic of k_edge's epilogue, value by value in the same order, without branches and without
    // exec-masked stores (either would end the basic block and with it the MFMA / VALU interleaving): lanes that have nothing to
    // store write to the dump part a.dump_part, both halves of the wavefront store where they hold the same value.
    struct Prev { int nseg, pid; uint32_t sw[4]; };
    float dot[16], wgt[16];
    float rowdot = 0.f;
    auto epi_silu_dot = [&](f32x16& acc, float wav, int r0, int r1) {   // SiLU of (part of) a column tile + its share of the row dots
#pragma unroll
        for (int r = r0; r < r1; ++r) {
            const float mv = HD_X6_SILU(acc[r]);
            acc[r] = mv;
            dot[r] = __builtin_fmaf(mv, wav, dot[r]);
        }
    };
    auto epi_reduce = [&]() {                                      // transpose-reduce: lanes 2s, 2s+1 hold the dot of row slot s
        float v8[8], v4[4], v2[2];
        const bool b4 = n & 16, b3 = n & 8, b2_ = n & 4, b1 = n & 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) v8[k] = (b4 ? dot[k + 8] : dot[k]) + __shfl_xor(b4 ? dot[k] : dot[k + 8], 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) v4[k] = (b3 ? v8[k + 4] : v8[k]) + __shfl_xor(b3 ? v8[k] : v8[k + 4], 8);
#pragma unroll
        for (int k = 0; k < 2; ++k) v2[k] = (b2_ ? v4[k + 2] : v4[k]) + __shfl_xor(b2_ ? v4[k] : v4[k + 2], 4);
        float v = (b1 ? v2[1] : v2[0]) + __shfl_xor(b1 ? v2[0] : v2[1], 2);
        rowdot = v + __shfl_xor(v, 1);
    };
    auto seg_of = [&](const Prev& p, int r) -> int { return (p.sw[r >> 2] >> (8 * (r & 3))) & 255; };
    auto epi_gate = [&](const Prev& p) {                           // GCL: attention weight of each of the lane's 16 rows
        const float sg = sigmoid_f(rowdot + att_bias);
        const float att_mine = a.attention ? sg : 1.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) wgt[r] = __shfl(att_mine, (lane & 32) | (2 * r));
    };
    // per-node sum of segment s: k_edge's select form (a NaN row stays inside its own segment; for finite rows the same
    // bits as its masked form, where the other rows contribute fma(0, m, sum) = sum)
    float sums[NCT];
    auto seg_col = [&](f32x16& acc, const Prev& p, int s, int ct) {        // one column tile of the sum of segment s
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum = (seg_of(p, r) == s) ? __builtin_fmaf(wgt[r], acc[r], sum) : sum;
        sums[ct] = sum;
    };
    auto seg_store = [&](const Prev& p, int s, int c4) {                   // four column tiles: add the wavefront's halves, store
        const int ps = __builtin_amdgcn_readlane(p.pid, s & 31);
        float* dst = a.part + (size_t)(s < p.nseg ? ps : a.dump_part) * H + n;
        float q4[4] = {sums[c4], sums[c4 + 1], sums[c4 + 2], sums[c4 + 3]};
        xhalf_sum4(q4);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[32 * (c4 + k)] = q4[k];
    };
    auto epi_segment = [&](f32x16 (&acc)[NCT], const Prev& p, int s) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) seg_col(acc[ct], p, s, ct);
        seg_store(p, s, 0);
        seg_store(p, s, 4);
    };
    auto epi_coord_a = [&](int par) {                              // phi per row -> tanh -> scale the unit directions in place
        float* sc = scr_w + par * 144;
        const int my_slot = (n >> 1) & 15;
        sc[(my_slot & 3) + 8 * (my_slot >> 2) + 4 * hh] = rowdot;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const float phi = sc[n];
        const float th = tanhf(phi) * a.coords_range;
        const float s = a.use_tanh ? th : phi;
        const float tx = sc[32 + n * 3 + 0] * s, ty = sc[32 + n * 3 + 1] * s, tz = sc[32 + n * 3 + 2] * s;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        sc[32 + n * 3 + 0] = tx; sc[32 + n * 3 + 1] = ty; sc[32 + n * 3 + 2] = tz;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto epi_coord_b = [&](const Prev& p, int par) {               // lane s sums the rows of segment s, in row order
        const float* sc = scr_w + par * 144;
        const u32x4 sb0 = *reinterpret_cast<const u32x4*>(sc + 128), sb1 = *reinterpret_cast<const u32x4*>(sc + 132);
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {                              // 4 rows = 12 floats = 3 quads
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(sc + 32 + 12 * g);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(sc + 36 + 12 * g);
            const f32x4 t2 = *reinterpret_cast<const f32x4*>(sc + 40 + 12 * g);
            const float tv[12] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3]};
            const uint32_t w = g < 4 ? sb0[g & 3] : sb1[g & 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool m = (int)((w >> (8 * k)) & 255) == lane;
                sx = m ? sx + tv[3 * k] : sx; sy = m ? sy + tv[3 * k + 1] : sy; sz = m ? sz + tv[3 * k + 2] : sz;
            }
        }
        *reinterpret_cast<f32x4*>(a.part + (size_t)(lane < p.nseg ? p.pid : a.dump_part) * 4) = f32x4{sx, sy, sz, 0.f};
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto prev_of = [&](const Tile& t, int par) {
        Prev p;
        p.nseg = t.nseg; p.pid = t.pid;
        const uint32_t* seg_s = reinterpret_cast<const uint32_t*>(scr_w + par * 144 + 128);
#pragma unroll
        for (int q = 0; q < 4; ++q) p.sw[q] = seg_s[2 * q + hh];
        return p;
    };
    auto load_x = [&](const Tile& t, f32x4& xi, f32x4& xj, f32x4& yi, f32x4& yj) {
        xi = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)t.ni * 4);
        xj = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)t.nj * 4);
        yi = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)t.ni * 4);
        yj = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)t.nj * 4);
    };
    // four pre-activations of quad u of K chunk c (k = 16c + 8hh + 4u + j) of tile t
    auto pre_quad = [&](float (&pre)[4], const Tile& t, const f32x4 qa, const f32x4 qb, const f32x4 wr4, const f32x4 wd4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float p = qa[j] + qb[j];
            p = __builtin_fmaf(t.radial, wr4[j], p);
            pre[j] = __builtin_fmaf(t.d0, wd4[j], p);
        }
    };

    // ---- first tile: the only exposed prologue
    Tile cur, nxt;
    f32x4 nxi, nxj, nyi, nyj;
    tile_meta(cur, wt_lo);
    issue_chunk(0, 0);
    issue_chunk(1, 1);
    load_x(cur, nxi, nxj, nyi, nyj);
    tile_geom(cur, nxi, nxj, nyi, nyj, 0);
    rows_issue(cur, 0, 0); rows_issue(cur, 0, 1);
    rows_issue(cur, 1, 0); rows_issue(cur, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // w_r / w_d / b2 / wa staged, W2 chunks 0, 1 and the rows of chunks 0, 1 landed
    u32x4 XA[3], XB[3];                    // head / middle / tail operand dwords of the even / odd K chunks
    X6Aux ax;
    ax.qa = ax.qb = ax.wr = ax.wd = f32x4{0.f, 0.f, 0.f, 0.f};
    ax.wavA = ax.wavB = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {          // operands of chunk 0
        f32x4 qa, qb;
        if (u == 0) x6p_read2<0, 1024>(qa, qb, rows_lane); else x6p_read2<2048, 3072>(qa, qb, rows_lane);
        lds_ready2(qa, qb);
        const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wrd_s + 8 * hh + 4 * u);
        const f32x4 wd4 = *reinterpret_cast<const f32x4*>(wrd_s + H + 8 * hh + 4 * u);
        float pre[4];
        pre_quad(pre, cur, qa, qb, wr4, wd4);
        uint32_t hi[2], mi[2], lo[2];
        bf16_split3(HD_X6_SILU(pre[0]), HD_X6_SILU(pre[1]), hi[0], mi[0], lo[0]);
        bf16_split3(HD_X6_SILU(pre[2]), HD_X6_SILU(pre[3]), hi[1], mi[1], lo[1]);
        XA[0][2 * u] = hi[0]; XA[0][2 * u + 1] = hi[1]; XA[1][2 * u] = mi[0]; XA[1][2 * u + 1] = mi[1]; XA[2][2 * u] = lo[0]; XA[2][2 * u + 1] = lo[1];
    }
    XB[0] = XB[1] = XB[2] = u32x4{0, 0, 0, 0};
    rows_issue(cur, 2, 0); rows_issue(cur, 2, 1);       // ring parity 0 is free again (chunk 0 consumed)
    // what stage 0 of chunk 0 needs: quad 0 of chunk 1's rows + w_r / w_d, wa of column tile 0
    x6p_read2<4096, 4096 + 1024>(ax.qa, ax.qb, rows_lane);
    x6p_read2<(16 + 0) * 4, (H + 16 + 0) * 4>(ax.wr, ax.wd, wrd_k);
    x6p_read1<(3 * H) * 4>(ax.wavA, wrd_n);

    f32x16 acc[NCT], accp[NCT];            // current tile / pending tile (epilogue riding)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const float b2v = wrd_s[2 * H + 32 * ct + n];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[ct][r] = b2v; accp[ct][r] = 0.f; }
    }
    Prev prev;                             // nothing pending yet: no segments, every store goes to the dump part
    prev.nseg = 0; prev.pid = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) prev.sw[q] = 0xffffffffu;
    int b0 = 0;                            // W2 buffer of chunk 0 of the current tile (chunk c lives in buffer (b0 + c) % 3; 16 % 3 = 1)
    int par = 0;                           // scratch parity of `cur`; the pending tile's is par ^ 1 (handed to the next tile in chunk 13)
    bf16x8 fA[2][4], fB[2][2];
    float pre[4], yv[4];

#pragma unroll 1
    for (int it = 0; it < wt_cnt; ++it) {
        const int wt_n = (wt_lo + it + 1 <= wt_last) ? wt_lo + it + 1 : wt_last;      // the last pass re-reads its own tile, unused
#pragma unroll
        for (int r = 0; r < 16; ++r) dot[r] = 0.f;
        static_for<0, NCH>([&](auto Cc) {
            constexpr int c = decltype(Cc)::value;
            const int buf = (b0 + c) % 3;
            // Chunk c (streamed two chunks ago) has landed in LDS and every wave is done with chunk c-1, whose buffer takes the
            // stream of chunk c+2 below.  VMEM operations younger than chunk c's stream: two row quads of chunk c-2's stages,
            // the stream of chunk c+1 and two row quads of chunk c-1 = 14 (next-tile loads only make the wait stricter).
            if constexpr (!(X6P_ABL & 2)) {
                if constexpr (X6P_ABL & 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");      // (measurement: no barrier)
                else asm volatile("s_waitcnt vmcnt(14)\n\ts_barrier" ::: "memory");
            }
            // next tile: tables (chunk 11), coordinates (12), geometry (13: the pending tile's scratch is free by then)
            if constexpr (c == 11) tile_meta(nxt, wt_n);
            if constexpr (c == 12) load_x(nxt, nxi, nxj, nyi, nyj);
            if constexpr (c == 13) tile_geom(nxt, nxi, nxj, nyi, nyj, par ^ 1);
            const unsigned wb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(wbuf + buf * CHF) + lane * 16;
            auto req_A = [&](auto G, bf16x8 (&f)[4]) {
                constexpr int g = decltype(G)::value;
                lds_read4<bf16x8, frag_off_x6<NCT>(2, 2 * g), frag_off_x6<NCT>(2, 2 * g + 1), frag_off_x6<NCT>(1, 2 * g),
                          frag_off_x6<NCT>(1, 2 * g + 1)>(f, wb_lds);
            };
            auto req_B = [&](auto G, bf16x8 (&f)[2]) {
                constexpr int g = decltype(G)::value;
                lds_read2f<bf16x8, frag_off_x6<NCT>(0, 2 * g), frag_off_x6<NCT>(0, 2 * g + 1)>(f, wb_lds);
            };
            req_A(std::integral_constant<int, 0>{}, fA[0]);
            req_B(std::integral_constant<int, 0>{}, fB[0]);
            if constexpr (!(X6P_ABL & 2) && !(X6P_ABL & 16)) issue_chunk((c + 2) & 15, (b0 + c + 2) % 3);
            u32x4(&X)[3] = (c & 1) ? XB : XA;
            u32x4(&Xn)[3] = (c & 1) ? XA : XB;
            const bf16x8 A_h = __builtin_bit_cast(bf16x8, X[0]), A_m = __builtin_bit_cast(bf16x8, X[1]), A_l = __builtin_bit_cast(bf16x8, X[2]);
            const Tile& gt = (c < 15) ? cur : nxt;                   // the tile whose chunk (c + 1) & 15 is generated here

            // LDS reads the riding work of tile stage S needs, issued one stage ahead (before that stage's fragment request)
            auto aux_issue = [&](auto Sc) {
                constexpr int S = decltype(Sc)::value & 127, sg = S & 7, cc = S >> 3;
                if constexpr (sg == 0 || sg == 4) {                 // quad u of chunk cc + 1: gathered rows + w_r / w_d
                    constexpr int u = sg >> 2, cn = (cc + 1) & 15, rp = cn & 1;
                    if constexpr (u == 0) x6p_read2_after_vm<12, rp * 4096, rp * 4096 + 1024>(ax.qa, ax.qb, rows_lane);
                    else x6p_read2<rp * 4096 + 2048, rp * 4096 + 3072>(ax.qa, ax.qb, rows_lane);
                    x6p_read2<(16 * cn + 4 * u) * 4, (H + 16 * cn + 4 * u) * 4>(ax.wr, ax.wd, wrd_k);
                }
                if constexpr (S % 12 == 0 && S < 96) {               // wa of column tile S / 12 (its SiLU slices start in stage S)
                    constexpr int k = S / 12;
                    if constexpr (k & 1) x6p_read1<(3 * H + 32 * k) * 4>(ax.wavB, wrd_n);
                    else x6p_read1<(3 * H + 32 * k) * 4>(ax.wavA, wrd_n);
                }
            };
            // the work riding under the six MFMAs of tile stage S
            auto ride = [&](auto Sc) {
                constexpr int S = decltype(Sc)::value, sg = S & 7, u = sg >> 2, j = sg & 3;
                // operands of the next chunk: one activation per stage, split three ways in pairs
                if constexpr (j == 0) pre_quad(pre, gt, ax.qa, ax.qb, ax.wr, ax.wd);
                yv[j] = HD_X6_SILU(pre[j]);
                if constexpr (j == 1 || j == 3) {
                    uint32_t hi, mi, lo;
                    bf16_split3(yv[j - 1], yv[j], hi, mi, lo);
                    Xn[0][2 * u + (j >> 1)] = hi; Xn[1][2 * u + (j >> 1)] = mi; Xn[2][2 * u + (j >> 1)] = lo;
                }
                // epilogue of the pending tile.  Stages 0-95: SiLU + row dots, four values per three stages (column tile k in
                // stages 12k .. 12k+11); 96: row reduction; 97: gate / coordinate scaling; 98: coordinate sums; 100-123: the
                // per-node sums of segments 0-2, a column tile per stage
                if constexpr (S < 96) {
                    constexpr int k3 = S / 3, pos = S % 3;
                    constexpr int v0 = 4 * k3 + (pos == 0 ? 0 : pos + 1), nv = pos == 0 ? 2 : 1;
                    constexpr int ct = v0 / 16, r0 = v0 % 16;
                    epi_silu_dot(accp[ct], (ct & 1) ? ax.wavB : ax.wavA, r0, r0 + nv);
                }
                if constexpr (S == 96) epi_reduce();
                if constexpr (S == 97) { if constexpr (COORD) epi_coord_a(par ^ 1); else epi_gate(prev); }
                if constexpr (COORD) {
                    if constexpr (S == 98) epi_coord_b(prev, par ^ 1);
                } else if constexpr (S >= 100 && S < 124) {
                    constexpr int s = (S - 100) >> 3, ct = (S - 100) & 7;
                    seg_col(accp[ct], prev, s, ct);
                    if constexpr (ct == 3) seg_store(prev, s, 0);
                    if constexpr (ct == 7) seg_store(prev, s, 4);
                }
                // rows of chunk c + 3 (of the next tile from chunk 13 on)
                if constexpr (sg == 1 || sg == 5) {
                    constexpr int cr = (c + 3) & 15, uq = sg >> 2;
                    if constexpr (c + 3 < 16) rows_issue(cur, cr, uq); else rows_issue(nxt, cr, uq);
                }
            };
            static_for<0, NP>([&](auto Gc) {
                constexpr int g = decltype(Gc)::value, c0 = 2 * g, c1 = 2 * g + 1;
                constexpr int SA = 8 * c + 2 * g, SB = SA + 1;
                bf16x8(&a4)[4] = fA[g & 1];
                bf16x8(&b2f)[2] = fB[g & 1];
                // ---- stage A: tail and middle fragments (h*L, h*M, m*M)
                x6p_wait4<2>(a4, ax);                   // behind this pair's A request and this stage's aux: its B request (2 reads)
                aux_issue(std::integral_constant<int, SA + 1>{});
                if constexpr (g + 1 < NP) req_A(std::integral_constant<int, g + 1>{}, fA[(g + 1) & 1]);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, a4[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, a4[1], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, a4[2], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, a4[3], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, a4[2], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, a4[3], acc[c1], 0, 0, 0);
                if constexpr (!(X6P_ABL & 1)) ride(std::integral_constant<int, SA>{});
#pragma unroll
                for (int k = 0; k < ((X6P_ABL & 4) ? 0 : 6); ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- stage B: head fragments (h*H, m*H, l*H)
                if constexpr (g + 1 < NP) x6p_wait2<4>(b2f, ax); else x6p_wait2<0>(b2f, ax);
                aux_issue(std::integral_constant<int, SB + 1>{});
                if constexpr (g + 1 < NP) req_B(std::integral_constant<int, g + 1>{}, fB[(g + 1) & 1]);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, b2f[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h, b2f[1], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, b2f[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_m, b2f[1], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_l, b2f[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_l, b2f[1], acc[c1], 0, 0, 0);
                if constexpr (X6P_ABL & 1) { if constexpr (g == 3) { Xn[0] = X[0]; Xn[1] = X[1]; Xn[2] = X[2]; } } else ride(std::integral_constant<int, SB>{});
#pragma unroll
                for (int k = 0; k < ((X6P_ABL & 4) ? 0 : 6); ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if constexpr (!COORD) {
            for (int s = 3; s < prev.nseg; ++s) epi_segment(accp, prev, s);       // rare: tail tiles shared by several molecules
        }
        // rotate: the finished tile becomes the pending one
        prev = prev_of(cur, par);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            accp[ct] = acc[ct];
            const float b2v = wrd_s[2 * H + 32 * ct + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = b2v;
        }
        cur = nxt;
        par ^= 1;
        b0 = (b0 + 1) % 3;
    }

    // ---- flush: epilogue of the last tile
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" : "+v"(ax.qa), "+v"(ax.qb), "+v"(ax.wr), "+v"(ax.wd), "+v"(ax.wavA), "+v"(ax.wavB) : : "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) dot[r] = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) epi_silu_dot(accp[ct], wrd_s[3 * H + 32 * ct + n], 0, 16);
    epi_reduce();
    if constexpr (COORD) {
        epi_coord_a(par ^ 1);
        epi_coord_b(prev, par ^ 1);
    } else {
        epi_gate(prev);
        for (int s = 0; s < prev.nseg; ++s) epi_segment(accp, prev, s);
    }
}
