// Experimental one-wavefront-per-SIMD edge kernel (k_edge_p, HD_EDGE_PIPE=1).  Included through kernels.hpp.
#pragma once
#include "k_edge.hpp"

// ----------------------------------------------------------------------------- pipelined edge kernel (bf16x3, H >= 256)
// EXPERIMENTAL, off by default (HD_EDGE_PIPE=1 selects it).  Measured on MI355X at B=256, N=30: 124 us (GCL) / 135 us
// (coord) against 103 us for k_edge.  Per-chunk stamps (scratch/edge_ptrace.py): 3.3-3.5 k cycles per 48-MFMA chunk
// (MFMA floor 1.5 k) and 7.7 k for stage B: with one wavefront per SIMD every instruction costs an issue slot of ~4
// cycles (~500 instructions per chunk) and every latency is exposed, while two co-resident wavefronts of k_edge
// share the SIMD's issue ports.  Kept as the starting point for a hand-scheduled version.
// Same arithmetic as k_edge<H, COORD, 1>, arranged for ONE wavefront per SIMD (512 registers) and persistent
// workgroups, because on gfx950 VALU work only overlaps matrix work when both sit in the same wavefront's
// instruction stream (scratch/mb/phased.hip: an MFMA-streaming wavefront and a VALU-streaming one on the same
// SIMD serialise; scratch/mb/coissue.hip: ~5 plain VALU issues per MFMA are free inside one wavefront):
//   * a wavefront walks its tiles; while the MFMAs of tile t run, its VALU slots carry (a) the operand generation
//     of tile t's next chunk as before and (b) stage A of tile t-1's epilogue - SiLU + attention/coordinate dot of
//     column tile c during K chunk c (two accumulator sets, 256 registers);
//   * stage B of tile t-1 (row-dot reduction, sigmoid / tanh, per-node sums, stores) runs between the chunk loops;
//   * the next tile's metadata, coordinates, first AB rows and first operand chunk are fetched / built inside the
//     last chunks of tile t, so a tile has no prologue of its own;
//   * the W2 stream runs two chunks ahead through three LDS buffers (one workgroup per CU) and never stops
//     between tiles.
// vmcnt bookkeeping (loads return in issue order; G = stream pieces per wave and chunk): the stream for chunk
// g is issued at the top of chunk g-2, every chunk issues 8 AB row gathers after it, so at the top of chunk g
// "at most 16 + G outstanding" retires the stream of chunk g, and before the first use of a gathered quad
// "at most 4 + G outstanding" retires the quad fetched one group ahead.  Compiler-visible loads (next tile's metadata) only ever
// make these waits stricter.

template <int H, bool COORD, bool TRACE = false>
__global__ __launch_bounds__(256, 1) void k_edge_p(EdgeArgs a) {
    long long tst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // TRACE: cycle stamps of the second tile

    constexpr int NCT = H / 32, NCH = H / 32, CHF = 32 * H, GLW = CHF / (4 * 256), NBUF = 3;
    static_assert(NCH >= 8, "the cross-tile prefetch schedule needs at least 8 K chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wbuf = smem;                                           // [NBUF][CHF]
    __shared__ __attribute__((aligned(16))) float wrd_s[4 * H];   // [w_r | w_d | b2 | wa]
    float* rows_all = smem + NBUF * CHF;                          // per wave: 8 slots x 1 KiB of gathered AB rows
    float* scratch = rows_all + 4 * 2048;                         // per wave: 32 phi + 96 trans + 2 x 8 seg words

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    float* my_scr = scratch + wave * 144;
    uint32_t* seg_s = reinterpret_cast<uint32_t*>(my_scr + 128);  // [2][8]
    const unsigned rows_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(rows_all + wave * 2048);
    const unsigned rows_lane = rows_lds + lane * 16;             // this lane's 16 bytes inside a slot

    int wt_first, wt_count, wt_step;
    {
        const int bid = blockIdx.x, G = gridDim.x, nwt = a.n_wg;
        const int xcd = bid & 7, slot = bid >> 3;
        const int q = nwt >> 3, r = nwt & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        wt_step = (G - xcd + 7) >> 3;
        wt_first = start + slot;
        wt_count = slot < len ? (len - slot + wt_step - 1) / wt_step : 0;
    }
    if (wt_count == 0) return;

    for (int k = tid; k < 2 * H; k += 256) wrd_s[k] = a.wrd[k];
    for (int k = tid; k < H; k += 256) { wrd_s[2 * H + k] = a.b2[k]; wrd_s[3 * H + k] = a.wa[k]; }
    auto issue_chunk = [&](int c, int buf) {
        const float* src = a.W2img + (size_t)c * CHF + wave * (GLW * 256);
        // opaque here: in the fully unrolled chunk loop hipcc otherwise hoists all NCH x GLW source addresses (a
        // 64-bit VGPR pair each) out of the tile loop and spills them
        asm volatile("" : "+s"(src));
        float* dst = wbuf + buf * CHF + wave * (GLW * 256);
#pragma unroll
        for (int u = 0; u < GLW; ++u) glds16(src + u * 256 + lane * 4, dst + u * 256);
    };
    issue_chunk(0, 0);
    issue_chunk(1, 1);

    // ---- per-tile state: geometry of this lane's edge row, row pointers, segment bookkeeping
    struct Tile {                            // 64 bytes, no padding (a padded tail is copied through scratch memory)
        const float* Arow;
        const float* Brow;
        int ni, nj, pbase, nseg, ok;
        uint32_t segb;
        float radial, d0, ux, uy, uz, spare; // u = (x_i - x_j) / (|x_i - x_j| + norm_constant) * valid   (COORD)
    };
    // No branches anywhere in the tile body (a branch splits it into basic blocks and hipcc then sinks the operand
    // generation of the chunk before the branch into the chunk after it).  Every tile index a wave can reach is
    // inside the padded edge tables: padding rows carry eseg = 255 and padding tiles nseg = 0.
    auto tile_meta = [&](Tile& t, int tile) {
        t.spare = 0.f;
        t.ok = tile < a.n_tiles;
        const int e = tile * 32 + n;
        t.ni = a.ei[e]; t.nj = a.ej[e]; t.segb = a.eseg[e];
        t.pbase = a.tile_pbase[tile]; t.nseg = a.tile_nseg[tile];
        t.Arow = a.AB + (size_t)t.ni * (2 * H) + 16 * hh;
        t.Brow = a.AB + (size_t)t.nj * (2 * H) + H + 16 * hh;
    };
    auto tile_geom = [&](Tile& t) {
        const f32x4 xi = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)t.ni * 4);
        const f32x4 xj = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)t.nj * 4);
        const f32x4 yi = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)t.ni * 4);
        const f32x4 yj = *reinterpret_cast<const f32x4*>(a.x0 + (size_t)t.nj * 4);
        const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
        t.radial = dx * dx + dy * dy + dz * dz;
        const float ex = yi[0] - yj[0], ey = yi[1] - yj[1], ez = yi[2] - yj[2];
        t.d0 = ex * ex + ey * ey + ez * ez;
        if constexpr (COORD) {
            const float inv = ((t.segb != 255) ? 1.0f : 0.0f) / (sqrtf(t.radial + 1e-8f) + a.norm_constant);
            t.ux = dx * inv; t.uy = dy * inv; t.uz = dz * inv;
        } else {
            t.ux = t.uy = t.uz = 0.f;
        }
    };

    // gathered AB rows of the chunk to be built next: quad U = (A_i[4], B_j[4]) in LDS slots 2U, 2U+1 of this wave
    auto rows_issue = [&](auto U, auto C, const Tile& t) {
        constexpr int u = decltype(U)::value, c = decltype(C)::value;
        vm_glds2(t.Arow + 32 * c + 4 * u, t.Brow + 32 * c + 4 * u, rows_lds + (2 * u) * 1024, rows_lds + (2 * u + 1) * 1024);
    };
    f32x4 qa[2], qb[2];                         // quad being consumed / quad fetched one group ahead
    auto make_pair = [&](const Tile& t, f32x2 av, f32x2 bv, f32x2 wr2, f32x2 wd2, uint32_t& hi, uint32_t& lo) {
        float y[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float pre = av[j] + bv[j];
            pre = __builtin_fmaf(t.radial, wr2[j], pre);
            pre = __builtin_fmaf(t.d0, wd2[j], pre);
            y[j] = silu_scaled(pre);
        }
        bf16_split2(y[0], y[1], hi, lo);
    };

    // ---- first tile: the only exposed prologue
    Tile cur, nxt;
    tile_meta(cur, wt_first * 4 + wave);
    tile_geom(cur);
    u32x4 P[2][2][2];                           // [chunk parity][head|tail][k-step]: operands of the current / next chunk
    static_for<0, 4>([&](auto U) { rows_issue(U, std::integral_constant<int, 0>{}, cur); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                            // w_r / w_d / b2 / wa staged; chunks 0 and 1 landed everywhere
    static_for<0, 4>([&](auto U) {
        constexpr int u = decltype(U)::value;
        lds_read2_after_vm<0, (2 * u) * 1024, (2 * u + 1) * 1024>(qa[0], qb[0], rows_lane);
        lds_ready2(qa[0], qb[0]);
        const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wrd_s + 16 * hh + 4 * u);
        const f32x4 wd4 = *reinterpret_cast<const f32x4*>(wrd_s + H + 16 * hh + 4 * u);
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
            uint32_t hi, lo;
            make_pair(cur, f32x2{qa[0][2 * j2], qa[0][2 * j2 + 1]}, f32x2{qb[0][2 * j2], qb[0][2 * j2 + 1]},
                      f32x2{wr4[2 * j2], wr4[2 * j2 + 1]}, f32x2{wd4[2 * j2], wd4[2 * j2 + 1]}, hi, lo);
            P[0][0][u >> 1][2 * (u & 1) + j2] = hi;
            P[0][1][u >> 1][2 * (u & 1) + j2] = lo;
        }
        rows_issue(U, std::integral_constant<int, 1>{}, cur);      // the slot pair is free again
    });
    // quad 0 of chunk 1: younger operations are its quads 1..3
    lds_read2_after_vm<6, 0, 1024>(qa[0], qb[0], rows_lane);

    // previous tile (stage A / B operate on it); starts out as an all-zero dummy
    f32x16 accp[NCT];
    float dot[16];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[ct][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dot[r] = 0.f;
    Tile prv = cur;
    int prv_tile = -1, prv_par = 0;

    // stage A of the epilogue for rows r0 .. r0+3 of column tile ct of the previous tile
    auto stage_a = [&](auto Ct, auto R0) {
        constexpr int ct = decltype(Ct)::value, r0 = decltype(R0)::value;
        const float wav = wrd_s[3 * H + 32 * ct + n];
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(accp[ct][r0 + j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = 1.0f + e[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) accp[ct][r0 + j] *= e[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) dot[r0 + j] = __builtin_fmaf(accp[ct][r0 + j], wav, dot[r0 + j]);
    };

    // stage B: row dots -> attention / coordinate head -> per-node sums of the previous tile.  Always executed
    // (nseg_b = 0 when there is nothing to store): inside a conditional block hipcc sinks ALL of stage A into it.
    auto stage_b = [&](int nseg_b) {
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
        const uint32_t* segw = seg_s + 8 * prv_par;
        if constexpr (!COORD) {
            uint32_t sw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) sw[q] = segw[2 * q + hh];
            float att_mine = 1.0f;
            if (a.attention) att_mine = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(rowdot + a.ba));
            float w[16];
            int sg[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sg[r] = (sw[r >> 2] >> (8 * (r & 3))) & 255;
                const float att = __shfl(att_mine, (lane & 32) | (2 * r));
                w[r] = (sg[r] != 255) ? att : 0.0f;
            }
            for (int s = 0; s < nseg_b; ++s) {
                float ws[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[r] = (sg[r] == s) ? w[r] : 0.0f;
                float* dst = a.part + (size_t)(prv.pbase + s) * H + n;
                float sums[NCT];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    float sum = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum = __builtin_fmaf(ws[r], accp[ct][r], sum);
                    sums[ct] = xhalf_sum(sum);
                }
                if (hh == 0) {
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) dst[32 * ct] = sums[ct];
                }
            }
        } else {
            if ((n & 1) == 0) my_scr[(my_slot & 3) + 8 * (my_slot >> 2) + 4 * hh] = rowdot;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (hh == 0) {
                const float phi = my_scr[n];
                const float sc = a.use_tanh ? tanhf(phi) * a.coords_range : phi;
                float* tr = my_scr + 32;
                tr[n * 3 + 0] = prv.ux * sc;
                tr[n * 3 + 1] = prv.uy * sc;
                tr[n * 3 + 2] = prv.uz * sc;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < nseg_b) {
                const uint8_t* sb = reinterpret_cast<const uint8_t*>(segw);
                const float* tr = my_scr + 32;
                float sx = 0.f, sy = 0.f, sz = 0.f;
                for (int rr = 0; rr < 32; ++rr) {
                    if (sb[rr] == lane) { sx += tr[rr * 3]; sy += tr[rr * 3 + 1]; sz += tr[rr * 3 + 2]; }
                }
                f32x4 o = {sx, sy, sz, 0.f};
                *reinterpret_cast<f32x4*>(a.part + (size_t)(prv.pbase + lane) * 4) = o;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };

    int buf = 0;                                // LDS buffer of the current chunk (stream position mod NBUF)
#pragma unroll 1
    for (int it = 0; it < wt_count; ++it) {
        const bool last_it = it + 1 == wt_count;
        const int par = it & 1;
        if (hh == 0) reinterpret_cast<uint8_t*>(seg_s + 8 * par)[n] = (uint8_t)cur.segb;
        nxt = cur;                              // placeholder when no tile follows (its results are never used)

        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const float b2v = wrd_s[2 * H + 32 * ct + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = b2v;
        }

        static_for<0, NCH>([&](auto Cc) {
            constexpr int c = decltype(Cc)::value;
            constexpr int cp = c & 1;                              // operand set of this chunk; the other one is being built
            constexpr int c1 = (c + 1) % NCH, c2 = (c + 2) % NCH;  // chunk built during this one / chunk whose rows are requested
            constexpr bool n1 = c + 1 >= NCH, n2 = c + 2 >= NCH;   // ... do they belong to the next tile?
            // chunk landed everywhere; everyone is done with the buffer the stream is about to overwrite
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TRACE) { if (it == 1) tst[c] = __builtin_readcyclecounter(); }
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "i"(16 + GLW) : "memory");
            {
                int nb = buf + 2; if (nb >= NBUF) nb -= NBUF;
                issue_chunk(c2, nb);
            }
            // the tile after this one (the current tile again when none follows: fetched, never used)
            if constexpr (c == NCH - 5) tile_meta(nxt, (wt_first + (last_it ? it : it + 1) * wt_step) * 4 + wave);
            if constexpr (c == NCH - 3) tile_geom(nxt);
            const Tile& t1 = n1 ? nxt : cur;
            const Tile& t2 = n2 ? nxt : cur;

            const float* wb = wbuf + buf * CHF;
            const unsigned wb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)wb + lane * 16;
            const float* wr_n = wrd_s + 32 * c1 + 16 * hh;
            const float* wd_n = wrd_s + H + 32 * c1 + 16 * hh;
            f32x4 wrq[2], wdq[2];
            wrq[0] = *reinterpret_cast<const f32x4*>(wr_n);
            wdq[0] = *reinterpret_cast<const f32x4*>(wd_n);
            bf16x8 f0[4], f1[4];
            lds_read4<bf16x8, frag_off_bf<NCT>(0, 0), frag_off_bf<NCT>(0, 1), frag_off_bf<NCT>(1, 0), frag_off_bf<NCT>(1, 1)>(f0, wb_lds);
            constexpr int NG = NCT;
            static_for<0, NG>([&](auto Gc) {
                constexpr int g = decltype(Gc)::value;
                bf16x8(&fc)[4] = (g & 1) ? f1 : f0;
                bf16x8(&fn)[4] = (g & 1) ? f0 : f1;
                lds_wait4<0>(fc);
                if constexpr (g + 1 < NG) {
                    constexpr int u = 2 * (g + 1);
                    lds_read4<bf16x8, frag_off_bf<NCT>(u, 0), frag_off_bf<NCT>(u, 1), frag_off_bf<NCT>(u + 1, 0),
                              frag_off_bf<NCT>(u + 1, 1)>(fn, wb_lds);
                }
                constexpr int NGP = NG / 2;
                static_assert(NGP == 4, "one AB quad per producing group");
                if constexpr (g < NGP) {
                    // group g builds the two operand pairs of quad g of chunk c1; the quad was fetched from LDS one
                    // group earlier (lgkmcnt(0) above covers it) and quad g+1 is fetched now.  Operations younger
                    // than quad g+1's gather: its quads g+2..3, this chunk's stream pieces, the quads 0..g-1 of
                    // chunk c2 issued so far  =  4 + GLW.
                    constexpr int u = g;
                    lds_ready2(qa[u & 1], qb[u & 1]);
                    if constexpr (u + 1 < 4) {
                        lds_read2_after_vm<4 + GLW, (2 * (u + 1)) * 1024, (2 * (u + 1) + 1) * 1024>(qa[(u + 1) & 1], qb[(u + 1) & 1], rows_lane);
                        wrq[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(wr_n + 4 * (u + 1));
                        wdq[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(wd_n + 4 * (u + 1));
                    }
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int pi = 2 * u + j2;
                        uint32_t hi, lo;
                        make_pair(t1, f32x2{qa[u & 1][2 * j2], qa[u & 1][2 * j2 + 1]}, f32x2{qb[u & 1][2 * j2], qb[u & 1][2 * j2 + 1]},
                                  f32x2{wrq[u & 1][2 * j2], wrq[u & 1][2 * j2 + 1]},
                                  f32x2{wdq[u & 1][2 * j2], wdq[u & 1][2 * j2 + 1]}, hi, lo);
                        P[cp ^ 1][0][pi >> 2][pi & 3] = hi;
                        P[cp ^ 1][1][pi >> 2][pi & 3] = lo;
                    }
                    rows_issue(std::integral_constant<int, u>{}, std::integral_constant<int, c2>{}, t2);   // slot pair free again
                } else {
                    // quad 0 of the chunk after next has been in flight since group 0: fetch it for the next chunk's
                    // group 0 (younger operations: its quads 1..3)
                    if constexpr (g == NG - 1) lds_read2_after_vm<6, 0, 1024>(qa[0], qb[0], rows_lane);
                    // stage A of the previous tile's epilogue: column tile c, four rows per group
                    stage_a(std::integral_constant<int, c>{}, std::integral_constant<int, 4 * (g - NGP)>{});
                }
                constexpr int u0 = 2 * g, u1 = 2 * g + 1;
                constexpr int s0 = u0 / NCT, c0 = u0 % NCT, s1 = u1 / NCT, cc1 = u1 % NCT;
                const bf16x8 A_h0 = __builtin_bit_cast(bf16x8, P[cp][0][s0]), A_l0 = __builtin_bit_cast(bf16x8, P[cp][1][s0]);
                const bf16x8 A_h1 = __builtin_bit_cast(bf16x8, P[cp][0][s1]), A_l1 = __builtin_bit_cast(bf16x8, P[cp][1][s1]);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h0, fc[0], acc[c0], 0, 0, 0);
                acc[cc1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h1, fc[2], acc[cc1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_l0, fc[0], acc[c0], 0, 0, 0);
                acc[cc1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_l1, fc[2], acc[cc1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h0, fc[1], acc[c0], 0, 0, 0);
                acc[cc1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_h1, fc[3], acc[cc1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 6; ++k) {                // interleave: 1 MFMA, then up to 5 VALU
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
                // the tile body is one straight-line block: without a fence per group hipcc piles the VALU work of
                // several chunks into a few of them and leaves the others as bare MFMA runs
                __builtin_amdgcn_sched_barrier(0);
            });
            ++buf; if (buf >= NBUF) buf -= NBUF;
        });

        // tile `it` is accumulated; finish the one before it, then rotate
        if constexpr (TRACE) { if (it == 1) tst[8] = __builtin_readcyclecounter(); }
        stage_b((prv_tile >= 0 && prv.ok) ? prv.nseg : 0);
        if constexpr (TRACE) { if (it == 1) tst[9] = __builtin_readcyclecounter(); }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) accp[ct] = acc[ct];
#pragma unroll
        for (int r = 0; r < 16; ++r) dot[r] = 0.f;
        prv = cur; prv_tile = it; prv_par = par;
        cur = nxt;
        if constexpr (TRACE) { if (it == 1) tst[10] = __builtin_readcyclecounter(); }
    }
    if constexpr (TRACE) {
        if (lane == 0) {
            long long* t = a.trace + ((size_t)blockIdx.x * 4 + wave) * 12;
            for (int k = 0; k < 11; ++k) t[k] = tst[k];
            t[11] = wt_count;
        }
    }
    // drain the stream and the unused last gathers, then finish the last tile without overlap
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_ready2(qa[0], qb[0]);
    static_for<0, NCT>([&](auto Ct) {
        static_for<0, 4>([&](auto Q) { stage_a(Ct, std::integral_constant<int, 4 * decltype(Q)::value>{}); });
    });
    stage_b(prv.ok ? prv.nseg : 0);
}
