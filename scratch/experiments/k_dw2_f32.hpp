// Exact-fp32 twin of k_dw2_x6 (hierdiff_amd/csrc/k_dw2.hpp), round 4: whole H x H result per slab, operands read once.
// Measured 251 us per layer against 298 us for the tiled split-K product of k_tgemm (+ a 64 MB slab reduction instead of 8 MB):
// no change of the training step (32.0 vs 31.8 ms, inside the box-to-box spread).  Not in the library.
#pragma once
#include "../../hierdiff_amd/csrc/k_dw2.hpp"

// The same product in EXACT fp32 (v_mfma_f32_32x32x2_f32) - the default arithmetic of the training path.  k_tgemm cuts the 256 x 256
// result into eight 64 x 128 tiles per K slab, so every operand row is read three times (1.4 GB of L2 / HBM reads per layer at
// B = 256): 298 us for 29 GFLOP.  Here, as in k_dw2_x6, one workgroup owns the whole H x H result of its slab of edge rows and reads
// every row once; what is left is the MFMA time (8 wavefronts x (H/128 x H/64) accumulators: 128 MFMAs per wavefront and 32-row chunk).
// LDS: both operands transposed on the write into [column][k] tiles of 32 k (row stride 36 floats, the eight k quads of a row
// XOR-permuted by (column >> 4) & 3 - k_tgemm's scheme: transposing ds_write_b32 2-way, fragment ds_read_b128 conflict-free),
// double-buffered: chunk c+1 is stored and chunk c+2 requested while chunk c's MFMAs run; one barrier per chunk.
template <int H>
constexpr int dw2f_lds_bytes() { return 2 * 2 * H * 36 * 4; }

template <int H>
__global__ __launch_bounds__(512, 2) void k_dw2_f32(Dw2Args a) {
    static_assert(H % 128 == 0, "wave grid 4 x 2 of 32 x 32 tiles");
    constexpr int MT = H / 128, NT = H / 64;
    constexpr int Q = H / 4;                            // float4 per operand row
    constexpr int RPP = 512 / Q;                        // rows (k) per pass
    constexpr int NPASS = 32 / RPP;
    constexpr int LDK = 36;
    constexpr int TILE = H * LDK;                       // floats per operand tile
    extern __shared__ __attribute__((aligned(16))) char lds_f[];
    float* base = reinterpret_cast<float*>(lds_f);      // [buf][G | P][H][LDK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int hh = lane >> 5, n = lane & 31;
    const int z = blockIdx.x;
    const int kbeg = z * a.kslab, kend = min(a.rows, kbeg + a.kslab);
    const int nchunk = (kend - kbeg) / 32;
    const int c4 = tid % Q, k0 = tid / Q;

    f32x4 gr[NPASS], pr[NPASS];
    auto load_chunk = [&](int c) {
        const int cc = c < nchunk ? c : nchunk - 1;     // past the slab: re-read the last chunk (never stored), no branch around loads
        const size_t r0 = (size_t)(kbeg + 32 * cc);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            gr[p] = *reinterpret_cast<const f32x4*>(a.G + (r0 + k0 + RPP * p) * H + 4 * c4);
            pr[p] = *reinterpret_cast<const f32x4*>(a.P + (r0 + k0 + RPP * p) * H + 4 * c4);
        }
    };
    auto sw = [](int row) { return (row >> 4) & 3; };
    auto store_chunk = [&](int buf) {
        float* gt = base + (size_t)buf * 2 * TILE;
        float* pt = gt + TILE;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int k = k0 + RPP * p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 4 * c4 + j;
                const int off = row * LDK + 4 * ((k >> 2) ^ sw(row)) + (k & 3);
                gt[off] = gr[p][j];
                pt[off] = pr[p][j];
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    if (nchunk > 0) {
        load_chunk(0);
        store_chunk(0);
        load_chunk(1);
        __syncthreads();
    }
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) store_chunk(buf ^ 1);       // chunk c+1 (requested during chunk c-1) into the buffer chunk c-1 used
        load_chunk(c + 2);                              // in flight under the MFMAs below
        const float* gt = base + (size_t)buf * 2 * TILE;
        const float* pt = gt + TILE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = wr * (H / 4) + 32 * mt + n;
                af[mt] = *reinterpret_cast<const f32x4*>(gt + row * LDK + 4 * ((4 * hh + q) ^ sw(row)));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = wc * (H / 2) + 32 * nt + n;
                bf[nt] = *reinterpret_cast<const f32x4*>(pt + row * LDK + 4 * ((4 * hh + q) ^ sw(row)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][j], bf[nt][j], acc[mt][nt], 0, 0, 0);
        }
        __syncthreads();                                // chunk c+1 is complete in the other buffer; chunk c's readers are done
    }

    float* out = a.ws + (size_t)z * H * H;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * (H / 4) + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh;
                __builtin_nontemporal_store(acc[mt][nt][r], out + (size_t)row * H + wc * (H / 2) + 32 * nt + n);
            }
}
