// EXPERIMENT (round 2), NOT part of the library: kept for the record, with its measurements.
//   Result: bit-identical to k_edge (46 GPU parity / shard-equality tests green), but SLOWER: 305.6 us (GCL) / 285.2 us (COORD)
//   against k_edge's 265.6 us at B = 256, N = 30, H = 256.  Ablations (gpurun_out -> profiles/history/r02_f32p_experiment.log): without
//   the riding VALU work 223.5 us (the MFMA floor at the power-limited clock), without barrier / stream 296.9 us, without
//   the forced MFMA / VALU interleave 300.3 us.  I.e. the VALU work adds its FULL issue time to the MFMA time: inside one
//   wavefront nothing overlaps with v_mfma_f32_32x32x2_f32.  scratch/mb/coissue32.hip confirms it in isolation (same log):
//   one wave per SIMD, 28.4 ns per MFMA alone, +2.2 ns per v_fma_f32 placed behind it (2.3 ns is its cost alone); two waves
//   per SIMD with VGPR accumulators hide about half of the other wave's VALU time, with AGPR accumulators none.  The fp32
//   "matrix" instruction evidently runs on the SIMD's packed-fp32 datapath (32 FMA / cycle, the rate of v_pk_fma_f32), not
//   beside it, so the one-wave-per-SIMD software pipeline has nothing to hide work under; k_edge's two waves per SIMD are
//   the better arrangement for this instruction, and its time is MFMA time + roughly half the VALU time.
//   To build it again: include it from kernels.hpp, add `int dump_part` to EdgeArgs (= n_parts, part buffers one row larger)
//   and launch grid = min(n_wg, CUs) with (2 * 32 * 256 + 4 * 4096) * 4 bytes of dynamic LDS.
//
// Persistent, software-pipelined form of the exact-fp32 edge kernel (H = 256).  Included through kernels.hpp.
//
// Same arithmetic and the same bits as k_edge<H, COORD, 0> (same tile tables, same operation order per value), arranged
// for ONE wavefront per SIMD (512 registers) walking several tiles, because the trace of k_edge (scratch/edge_trace.py,
// DESIGN.md section 4) shows where its matrix-pipe idle time comes from: a wave-tile spends 26 % of its life in a
// VALU / latency-only prologue and epilogue, and the co-resident wavefront keeps the pipe only ~63 % busy meanwhile.
// With 64-cycle fp32 MFMAs one wavefront has 16 issue slots per MFMA, so here everything rides in the same instruction
// stream as the MFMAs of the current tile:
//   * the epilogue of tile t-1 (SiLU + attention / coordinate dot, row reduction, gate, per-node sums, stores), cut into
//     eight slices, one per K chunk of tile t, working on a second accumulator set;
//   * the next tile's metadata, coordinates and first AB rows (requested in chunks 3-6), so a tile has no prologue;
//   * the operand generation of the next K chunk, as before.
// AB rows travel global -> LDS by DMA (16 B per lane, lane-linear 1 KiB slots) and are read back with ds_read_b128: an
// inline-asm load with a VGPR destination is not safe in a kernel that also uses accumulation registers (hipcc may copy a
// still-in-flight destination), and a compiler-visible load next to the W2 stream costs a vmcnt(0) per use.
// VMEM bookkeeping (operations complete in issue order): every chunk issues, after its barrier, [next-tile loads in
// chunks 3/4] + the 8 stream pieces of the next chunk + 8 row-gather pieces for the chunk after next.  "At most 8
// outstanding" at the next barrier therefore retires everything but those row gathers; "at most 14 outstanding" before
// quad u of the next chunk's rows is read retires that quad (and is merely stricter in the chunks with extra loads).
#pragma once
#include "k_edge.hpp"

// LDS reads of the pipelined kernel are inline asm throughout the chunk loop (W2 fragments, gathered AB rows, w_r / w_d /
// wa values): each unit of 16 MFMAs starts with ONE lgkmcnt(0) that releases what the previous unit requested, and requests
// what the next unit needs, so every read has a full unit (1024 matrix-pipe cycles) to land.  A compiler-visible ds_read
// would be guarded by a compiler-placed lgkmcnt wait in the middle of the MFMA stream, which also waits for the asm reads
// issued before it.
#ifndef F32P_ABL
#define F32P_ABL 0          // measurement builds only (scratch/f32p_ablate.sh): 1 = no riding work, 2 = no barrier / W2 stream, 4 = no forced interleave
#endif
struct UnitAux { f32x4 qa, qb, wr, wd; float wav; };
template <unsigned O0, unsigned O1>
HD_DEVINL void lds_read2(f32x4& x, f32x4& y, unsigned addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(x), "=&v"(y) : "v"(addr), "i"(O0), "i"(O1));
}
template <unsigned O>
HD_DEVINL void lds_read1(float& x, unsigned addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(x) : "v"(addr), "i"(O)); }
HD_DEVINL void unit_wait(f32x4 (&f)[4], UnitAux& x) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(x.qa), "+v"(x.qb), "+v"(x.wr), "+v"(x.wd), "+v"(x.wav));
}

template <int H, bool COORD, int VPM = 6>
__global__ __launch_bounds__(256, 1) void k_edge_f32p(EdgeArgs a) {
    constexpr int NCT = H / 32, NCH = H / 32, CHF = 32 * H, GLW = CHF / (4 * 256);
    static_assert(NCH == 8 && NCT == 8, "the slice schedule is written for H = 256");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wbuf = smem;                                           // [2][CHF]
    __shared__ __attribute__((aligned(16))) float wrd_s[4 * H];   // [w_r | w_d | b2 | wa]
    float* rows_all = smem + 2 * CHF;                             // per wave: [2 chunk parities][8 slots] x 1 KiB (asm access only)
    // own LDS object, like wrd_s: a compiler-visible LDS access that may alias the destination of an in-flight
    // global_load_lds makes hipcc wait vmcnt(0), and these are read and written inside the chunk loop
    __shared__ __attribute__((aligned(16))) float scr_s[4 * 2 * 144];   // per wave, per tile parity: 32 phi + 96 trans + 8 seg words

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    float* scr_w = scr_s + wave * 2 * 144;
    const float att_bias = a.ba_ptr ? *a.ba_ptr : a.ba;
    const unsigned rows_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(rows_all + wave * 4096);
    const unsigned rows_lane = rows_lds + lane * 16;
    const unsigned wrd_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)wrd_s;
    const unsigned wrd_k = wrd_lds + 64 * hh, wrd_n = wrd_lds + 4 * n;      // this lane's k-slice of w_r / w_d; its column of b2 / wa

    // contiguous share of the workgroup-tiles for this (persistent) workgroup
    int wt_lo, wt_cnt;
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
        t.Arow = a.AB + (size_t)t.ni * (2 * H) + 16 * hh;
        t.Brow = a.AB + (size_t)t.nj * (2 * H) + H + 16 * hh;
    };
    // geometry of this lane's edge row; COORD: the unit direction goes to the wave's scratch of tile parity `par`
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
    // row gathers of K chunk c of tile t, quad u: A quad -> slot 2u, B quad -> slot 2u+1 of ring parity `rp`
    auto rows_issue = [&](const Tile& t, int c, int u, int rp) {
        vm_glds2(t.Arow + 32 * c + 4 * u, t.Brow + 32 * c + 4 * u, rows_lds + rp * 8192 + (2 * u) * 1024,
                 rows_lds + rp * 8192 + (2 * u + 1) * 1024);
    };
    float PcA[16], PcB[16];                                        // operands of the even / odd K chunks
    auto gen_math = [&](float (&Pc)[16], int u, const Tile& t, const f32x4 qa, const f32x4 qb, const f32x4 wr4, const f32x4 wd4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pre = qa[j] + qb[j];
            pre = __builtin_fmaf(t.radial, wr4[j], pre);
            pre = __builtin_fmaf(t.d0, wd4[j], pre);
            Pc[4 * u + j] = silu_f(pre);
        }
    };
    auto gen_quad = [&](float (&Pc)[16], auto U, auto RP, const Tile& t, int c) {   // prologue: Pc[4u..4u+3] of chunk c from ring parity RP
        constexpr int u = decltype(U)::value, rp = decltype(RP)::value;
        f32x4 qa, qb;
        lds_read2_after_vm<6 + GLW, rp * 8192 + (2 * u) * 1024, rp * 8192 + (2 * u + 1) * 1024>(qa, qb, rows_lane);
        lds_ready2(qa, qb);
        const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wrd_s + 32 * c + 16 * hh + 4 * u);
        const f32x4 wd4 = *reinterpret_cast<const f32x4*>(wrd_s + H + 32 * c + 16 * hh + 4 * u);
        gen_math(Pc, u, t, qa, qb, wr4, wd4);
    };
    // what unit u of chunk c reads from LDS besides its W2 fragments; requested one unit ahead
    auto aux_read = [&](auto Cc, auto Uc, UnitAux& x) {
        constexpr int c = decltype(Cc)::value, u = decltype(Uc)::value;
        if constexpr (c < 4) lds_read1<(3 * H + 32 * (2 * c + (u >> 2))) * 4>(x.wav, wrd_n);
        if constexpr (u & 1) {
            constexpr int q = u >> 1, cn = (c + 1) & 7, rp = cn & 1;
            lds_read2_after_vm<6 + GLW, rp * 8192 + (2 * q) * 1024, rp * 8192 + (2 * q + 1) * 1024>(x.qa, x.qb, rows_lane);
            lds_read2<(32 * cn + 4 * q) * 4, (H + 32 * cn + 4 * q) * 4>(x.wr, x.wd, wrd_k);
        }
    };

    // ---- epilogue pieces: the arithmetic of k_edge's epilogue, value by value in the same order, written without
    // branches and without exec-masked stores (either would end the basic block and with it the MFMA / VALU
    // interleaving): lanes that have nothing to store write to the dump part a.dump_part, both halves of the wavefront
    // store where they hold the same value.
    struct Prev { int nseg, pid; uint32_t sw[4]; };
    float dot[16], wgt[16];
    float rowdot = 0.f;
    auto epi_silu_dot = [&](f32x16& acc, float wav, int r0, int r1) {   // SiLU of (part of) a column tile + its share of the row dots
#pragma unroll
        for (int r = r0; r < r1; ++r) {
            const float mv = silu_f(acc[r]);
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

    // ---- first tile: the only exposed prologue
    Tile cur, nxt;
    f32x4 nxi, nxj, nyi, nyj;
    tile_meta(cur, wt_lo);
    issue_chunk(0, 0);
    load_x(cur, nxi, nxj, nyi, nyj);
    tile_geom(cur, nxi, nxj, nyi, nyj, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) rows_issue(cur, 0, u, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // w_r / w_d / b2 / wa staged, W2 chunk 0 and the rows of chunk 0 landed
    static_for<0, 4>([&](auto U) { gen_quad(PcA, U, std::integral_constant<int, 0>{}, cur, 0); });
#pragma unroll
    for (int u = 0; u < 4; ++u) rows_issue(cur, 1, u, 1);

    f32x16 acc[NCT], accp[NCT];            // current tile / previous tile (epilogue pending)
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
    UnitAux auxA, auxB;
    auxA.qa = auxA.qb = auxA.wr = auxA.wd = auxB.qa = auxB.qb = auxB.wr = auxB.wd = f32x4{0.f, 0.f, 0.f, 0.f};
    auxA.wav = auxB.wav = 0.f;
    aux_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, auxA);
    int par = 0;                           // scratch parity of `cur`; the pending tile's is par ^ 1, and it is handed on to
                                           // the next tile once the pending epilogue has read it (chunk 6)

#pragma unroll 1
    for (int it = 0; it < wt_cnt; ++it) {
        const int wt_n = (wt_lo + it + 1 <= wt_last) ? wt_lo + it + 1 : wt_last;      // the last pass re-reads its own tile, unused
#pragma unroll
        for (int r = 0; r < 16; ++r) dot[r] = 0.f;
        static_for<0, NCH>([&](auto Cc) {
            constexpr int c = decltype(Cc)::value;
            constexpr int buf = c & 1;
            // chunk c landed in LDS and every wave is done with the other buffer; the row gathers of one chunk may stay in flight
            if constexpr (!(F32P_ABL & 2)) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            // next tile: tables (chunk 3), coordinates (5), geometry (6).  Right behind the barrier, where nothing but row
            // gathers is in flight: hipcc guards the first use of a compiler-visible load with vmcnt(0) when LDS-DMA
            // operations are outstanding, which here costs nothing
            if constexpr (c == 3) tile_meta(nxt, wt_n);
            if constexpr (c == 5) load_x(nxt, nxi, nxj, nyi, nyj);
            if constexpr (c == 6) tile_geom(nxt, nxi, nxj, nyi, nyj, par ^ 1);
            const float* wb = wbuf + buf * CHF;
            const unsigned wb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)wb + lane * 16;
            constexpr int HC = 4, UPQ = NCT / HC, NU = 4 * UPQ;
            f32x4 f0[4], f1[4];
            auto read_unit = [&](auto U, f32x4 (&f)[4]) {
                constexpr int u = decltype(U)::value, q = u / UPQ, c0 = (u % UPQ) * HC;
                lds_read4<f32x4, frag_off_f32(q * NCT + c0), frag_off_f32(q * NCT + c0 + 1), frag_off_f32(q * NCT + c0 + 2),
                          frag_off_f32(q * NCT + c0 + 3)>(f, wb_lds);
            };
            read_unit(std::integral_constant<int, 0>{}, f0);
            if constexpr (!(F32P_ABL & 2)) issue_chunk((c + 1) & 7, buf ^ 1);
            float(&Pc)[16] = (c & 1) ? PcB : PcA;
            float(&Pn)[16] = (c & 1) ? PcA : PcB;
            // Work that rides under the MFMAs of unit u of this chunk.  Each unit is its own scheduling region (16 MFMAs,
            // 1024 matrix-pipe cycles, at most ~100 VALU instructions), inside which MFMA and VALU issue alternate; ax holds
            // what the unit needs from LDS (requested by the previous unit).
            auto ride = [&](auto Uc, const UnitAux& ax) {
                constexpr int u = decltype(Uc)::value;
                // operands of the next chunk, quad u / 2 (rows requested a chunk ago), then the rows of the chunk after next
                if constexpr (u & 1) gen_math(Pn, u >> 1, c < 7 ? cur : nxt, ax.qa, ax.qb, ax.wr, ax.wd);
                // epilogue of the pending tile: chunks 0-3 SiLU + row dots (4 rows of a column tile per unit), chunk 4 the
                // row reduction and the gate / coordinate scaling, chunks 5-7 the sums of segments 0-2 (a column tile per unit)
                if constexpr (c < 4) epi_silu_dot(accp[2 * c + (u >> 2)], ax.wav, 4 * (u & 3), 4 * (u & 3) + 4);
                if constexpr (c == 4 && u == 0) epi_reduce();
                if constexpr (c == 4 && u == 2) { if constexpr (COORD) epi_coord_a(par ^ 1); else epi_gate(prev); }
                if constexpr (COORD) {
                    if constexpr (c == 5 && u == 0) epi_coord_b(prev, par ^ 1);
                } else if constexpr (c >= 5) {
                    seg_col(accp[u], prev, c - 5, u);
                    if constexpr (u == 3) seg_store(prev, c - 5, 0);
                    if constexpr (u == 7) seg_store(prev, c - 5, 4);
                }
                if constexpr (u & 1) {
                    constexpr int q = u >> 1;
                    if constexpr (c < 6) rows_issue(cur, c + 2, q, c & 1);
                    else rows_issue(nxt, c - 6, q, c & 1);
                }
            };
            static_for<0, NU>([&](auto Uc) {
                constexpr int u = decltype(Uc)::value, q = u / UPQ, c0 = (u % UPQ) * HC;
                f32x4(&cf)[4] = (u & 1) ? f1 : f0;
                f32x4(&nf)[4] = (u & 1) ? f0 : f1;
                UnitAux& ax = (u & 1) ? auxB : auxA;
                UnitAux& an = (u & 1) ? auxA : auxB;
                unit_wait(cf, ax);
                if constexpr (u + 1 < NU) {
                    read_unit(std::integral_constant<int, u + 1>{}, nf);
                    aux_read(Cc, std::integral_constant<int, u + 1>{}, an);
                } else {
                    aux_read(std::integral_constant<int, (c + 1) & 7>{}, std::integral_constant<int, 0>{}, an);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < HC; ++ct)
                        acc[c0 + ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(Pc[4 * q + j], cf[ct][j], acc[c0 + ct], 0, 0, 0);
                if constexpr (F32P_ABL & 1) { if constexpr (u == 7) { for (int k = 0; k < 16; ++k) Pn[k] = Pc[k] + ax.wav; } }
                else ride(Uc, ax);
                // one MFMA (16 issue slots of matrix-pipe time), then a few VALU instructions of the riding work: left alone
                // hipcc issues the MFMAs back to back and the VALU work in blocks of 20-500 instructions
#pragma unroll
                for (int k = 0; k < ((F32P_ABL & 4) ? 0 : 4 * HC); ++k) {
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
    }

    // ---- flush: epilogue of the last tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
