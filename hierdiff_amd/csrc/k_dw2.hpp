// dW2 = G2^T P on the matrix cores proper (training, opt-in fp16x3 arithmetic): the one dense reduction over ALL edge rows of an
// edge layer's backward, dL/dW2[c][k] = sum_e G2[e][c] P[e][k] (hd_edge_layer_backward_s leaves G2 = dL/d(pre2) and P = SiLU(pre1)
// as [rows][H] arrays; reference: the autograd of edge_mlp.2 / coord_mlp.2, egnn_new.py:19-23, 83-86).  Included through kernels.hpp.
//
// In exact fp32 (k_tgemm) this product is 29 GFLOP per layer at H = 256, B = 256 on an instruction that runs at the vector rate.
// Here (k_dw2_f16, below) both operands are split two ways into FP16 pieces and the product is formed from three
// v_mfma_f32_32x32x16_f16 per k-step with fp32 accumulation.  One workgroup owns the WHOLE H x H result for a slab of edge rows
// (split-K: partial results in `ws`, streamed, added in a fixed order by k_dw2_reduce - deterministic), so each operand row is read
// exactly once:
//   * 32 edge rows per chunk: every thread loads float4 pieces of two consecutive rows (coalesced), splits the 2 x 4 values
//     and stores them as packed pairs into two LDS planes per operand, [column][k] with k contiguous (the MFMA wants 8
//     consecutive k of one column per lane; memory has the columns contiguous): row stride 80 B, the four 16-byte k groups of a
//     row XOR-permuted by (column >> 4) & 3, which makes the transposing ds_write_b32 2-way (free) instead of 8-way conflicted
//     and keeps the ds_read_b128 fragment reads conflict-free;
//   * 8 wavefronts as 4 (result rows) x 2 (result columns), H/128 x H/64 accumulators of 32 x 32 each;
//   * the next chunk's global loads are in flight under the current chunk's MFMAs (single LDS buffer, two barriers per chunk).
// (Rounds 4-5 carried a three-way bf16 twin of this kernel, k_dw2_x6; retired in round 6 with the bf16x6 mode.)
#pragma once
#include "common.hpp"

// dW2[m][n] = sum over the slabs' partial results, in a FIXED order: eight interleaved running sums (slab z goes to sum z % 8,
// ascending z) added as ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)) - eight loads in flight per thread instead of a
// dependent chain over up to 256 slabs, the same bits run to run.
__global__ __launch_bounds__(256) void k_dw2_reduce(const float* ws, float* C, int total, int N, int ldc, int nz) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= nz; z += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += __builtin_nontemporal_load(ws + (size_t)(z + k) * total + idx);
    }
    for (int k = 0; z + k < nz; ++k) s[k] += __builtin_nontemporal_load(ws + (size_t)(z + k) * total + idx);
    C[(size_t)(idx / N) * ldc + idx % N] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// ----------------------------------------------------------------------------- fp16x3 arithmetic of the training path (round 5)
// Device scalars of one edge layer's fp16 images, computed from the PARAMETERS themselves (training: they change every step):
//   scal[0] = 2^k, the power of two that puts max |W2| into [2^14, 2^15)  (pack_edge_w2_f16's rule),  scal[1] = 2^-k,
//   scal[2] = max |w_r|, scal[3] = max |w_d|  (the distance terms of the forward kernel's row bound).
// One workgroup of 1024 threads: H x H + 2 H values.
__global__ __launch_bounds__(1024) void k_f16_prep(const float* W2, const float* wrd, float* scal, int H) {
    __shared__ float red[3][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wm = 0.f, rm = 0.f, dm = 0.f;
    // (H >= 128: all of a thread's 4 - 16 loads in flight at once)
    for (int i0 = tid * 4; i0 < H * H; i0 += 4 * 4096) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u * 4096 < H * H) ? *reinterpret_cast<const f32x4*>(W2 + i0 + u * 4096) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float x = fabsf(v[u][j]); wm = (x > wm && x <= 3.4028234663852886e38f) ? x : wm; }      // finite values only
    }
    if (wrd) for (int i = tid; i < H; i += 1024) { rm = fmaxf(rm, fabsf(wrd[i])); dm = fmaxf(dm, fabsf(wrd[H + i])); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { wm = fmaxf(wm, __shfl_xor(wm, o)); rm = fmaxf(rm, __shfl_xor(rm, o)); dm = fmaxf(dm, __shfl_xor(dm, o)); }
    if (lane == 0) { red[0][wave] = wm; red[1][wave] = rm; red[2][wave] = dm; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) { wm = fmaxf(wm, red[0][w]); rm = fmaxf(rm, red[1][w]); dm = fmaxf(dm, red[2][w]); }
        // wm = m 2^ex with m in [0.5, 1)  =>  k = 15 - ex, clamped like the host packer
        int k = 0;
        if (wm > 0.f) {
            const int ex = (int)((__builtin_bit_cast(uint32_t, wm) >> 23) & 0xffu) - 126;       // subnormal maxima clamp below
            k = max(-100, min(100, 15 - ex));
        }
        scal[0] = __builtin_bit_cast(float, (uint32_t)(127 + k) << 23);
        scal[1] = __builtin_bit_cast(float, (uint32_t)(127 - k) << 23);
        scal[2] = rm; scal[3] = dm;
    }
}

// W [H][H] -> fp16 chunk image of the two-way edge kernels (pack_edge_w2_f16 on the device): per 32-wide K chunk
// [hi | lo][2 k-steps][H/32 column tiles][64 lanes][8 halves], lane (hh, n) element i = W[32 ct + n][32 c + 16 hh + 8 st + i] x 2^k -
// or, TRANS, of W^T.  One thread per (chunk, k-step, tile, lane): eight values, two 16-byte stores.
template <bool TRANS>
__global__ void k_pack_w2_f16(const float* W, const float* scal, f16x8* img, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int NCT = H / 32;
    if (idx >= (H / 32) * 2 * NCT * 64) return;
    const int lane = idx & 63, rest = idx >> 6;
    const int ct = rest % NCT, st = (rest / NCT) & 1, c = rest / (2 * NCT);
    const int col = 32 * ct + (lane & 31), k0 = 32 * c + 16 * (lane >> 5) + 8 * st;
    const float sw = scal[0];
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = (TRANS ? W[(size_t)(k0 + i) * H + col] : W[(size_t)col * H + k0 + i]) * sw;
        hi[i] = (_Float16)v;
        lo[i] = (_Float16)(v - (float)hi[i]);
    }
    const size_t blk = (size_t)c * 4 * NCT * 64;                     // f16x8 units per chunk: 2 pieces x 2 k-steps x NCT x 64
    img[blk + ((size_t)(0 * 2 + st) * NCT + ct) * 64 + lane] = hi;
    img[blk + ((size_t)(1 * 2 + st) * NCT + ct) * 64 + lane] = lo;
}

// The image of the backward stage (k_edge_bwd PREC 3): k_pack_w2_x6's geometry with two fp16 pieces - per 16-wide K chunk
// [hi | lo][H/32 column tiles][64 lanes][8 halves], lane (hh, n) element i = W[32 ct + n][16 c + 8 hh + i] x 2^k, or (TRANS) of W^T.
template <bool TRANS>
__global__ void k_pack_w2_f16c(const float* W, const float* scal, f16x8* img, int H) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;            // (c * NCT + ct) * 64 + lane
    const int NCT = H / 32;
    if (idx >= (H / 16) * NCT * 64) return;
    const int lane = idx & 63, ct = (idx >> 6) % NCT, c = (idx >> 6) / NCT;
    const int col = 32 * ct + (lane & 31), k0 = 16 * c + 8 * (lane >> 5);
    const float sw = scal[0];
    f16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = (TRANS ? W[(size_t)(k0 + i) * H + col] : W[(size_t)col * H + k0 + i]) * sw;
        hi[i] = (_Float16)v;
        lo[i] = (_Float16)(v - (float)hi[i]);
    }
    const size_t base = ((size_t)c * 2 * NCT + ct) * 64 + lane;      // f16x8 units: 2 pieces x NCT x 64 per chunk
    img[base] = hi;
    img[base + (size_t)NCT * 64] = lo;
}

// dW2 = G2^T P in fp16x3 arithmetic (structure: this file's header - one workgroup per slab of edge rows, [column][k] planes in LDS,
// 4 x 2 wavefronts): two FP16 planes per operand and three MFMAs per product (h*h, l*h, h*l; what is dropped is <= 2^-21 of a product).
// Both operands are ranged by ONE power of two each - G2 by 2^(14 - E(max |G2|)), P likewise: a sum over all edge rows is as exact as
// its largest terms are, an element 2^18 below the array's maximum still keeps 22 significant bits (head normal, tail >= the
// subnormal quantum 2^-24) and smaller ones lose bits in proportion to how little they matter.  The two maxima are per-workgroup
// maxima left by the backward stages (EdgeBwdArgs.g2wgmax / pwgmax), reduced here by every workgroup in its prologue (28 KB from L2:
// no extra launch, no atomics, deterministic).  The result leaves as acc x 2^-(a + b), exact.
struct Dw2F16Args {
    const float* G; const float* P; float* ws;
    const float* gmax; const float* pmax;       // [nmax] per-workgroup maxima of |G2|, |P|
    int rows, kslab, nmax;
};

template <int H>
constexpr int dw2_f16_lds_bytes() { return 2 * 2 * H * 80; }

template <int H>
__global__ __launch_bounds__(512, 2) void k_dw2_f16(Dw2F16Args a) {
    static_assert(H % 128 == 0, "wave grid 4 x 2 of 32 x 32 tiles");
    constexpr int MT = H / 128, NT = H / 64;
    constexpr int Q = H / 4;
    constexpr int NPASS = H / 128;
    constexpr int PLANE = H * 80;
    extern __shared__ __attribute__((aligned(16))) char lds_d[];
    char* gpl = lds_d;                                  // G planes hi, lo
    char* ppl = lds_d + 2 * PLANE;                      // P planes
    __shared__ float redm[2][8];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int kg = lane >> 5, i = lane & 31;
    const int z = blockIdx.x;
    const int kbeg = z * a.kslab, kend = min(a.rows, kbeg + a.kslab);
    const int nchunk = (kend - kbeg) / 32;

    const int c4 = tid % Q, kp0 = tid / Q;
    constexpr int KPP = 512 / Q;
    f32x4 gr[NPASS][2], pr[NPASS][2];
    auto load_chunk = [&](int c) {
        const size_t r0 = (size_t)(kbeg + 32 * c);
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const size_t row = r0 + 2 * (kp0 + KPP * p);
            gr[p][0] = *reinterpret_cast<const f32x4*>(a.G + row * H + 4 * c4);
            gr[p][1] = *reinterpret_cast<const f32x4*>(a.G + (row + 1) * H + 4 * c4);
            pr[p][0] = *reinterpret_cast<const f32x4*>(a.P + row * H + 4 * c4);
            pr[p][1] = *reinterpret_cast<const f32x4*>(a.P + (row + 1) * H + 4 * c4);
        }
    };
    if (nchunk > 0) load_chunk(0);                      // in flight under the reduction of the maxima

    // global ranges of the two operands
    float gs, ps, unscale;
    {
        float gm = 0.f, pm = 0.f;
        for (int k = tid; k < a.nmax; k += 512) { gm = fmaxf(gm, a.gmax[k]); pm = fmaxf(pm, a.pmax[k]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { gm = fmaxf(gm, __shfl_xor(gm, o)); pm = fmaxf(pm, __shfl_xor(pm, o)); }
        if (lane == 0) { redm[0][wave] = gm; redm[1][wave] = pm; }
        __syncthreads();
        gm = redm[0][0]; pm = redm[1][0];
#pragma unroll
        for (int w = 1; w < 8; ++w) { gm = fmaxf(gm, redm[0][w]); pm = fmaxf(pm, redm[1][w]); }
        // x in [2^(eb-127), 2^(eb-126)): scale 2^(141 - eb) puts it into [2^14, 2^15); all-zero or subnormal maxima scale by 2^100 at most
        auto scale_of = [](float m, float& inv) {
            const int eb = (int)((__builtin_bit_cast(uint32_t, fminf(m, 3.0e38f)) >> 23) & 0xffu);
            const int k = min(100, 141 - max(eb, 1));
            inv = __builtin_bit_cast(float, (uint32_t)(127 - k) << 23);
            return __builtin_bit_cast(float, (uint32_t)(127 + k) << 23);
        };
        float gi, pi;
        gs = scale_of(gm, gi); ps = scale_of(pm, pi);
        unscale = gi * pi;                              // may underflow to a subnormal only when both arrays are ~2^-100: the result is then 0 anyway
    }

    auto slot = [](int col, int kp) { return col * 80 + (((kp >> 2) ^ ((col >> 4) & 3)) << 4) + ((kp & 3) << 2); };
    auto store_chunk = [&]() {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int kp = kp0 + KPP * p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = slot(4 * c4 + j, kp);
                uint32_t h, l;
                f16_split2(gr[p][0][j] * gs, gr[p][1][j] * gs, h, l);
                *reinterpret_cast<uint32_t*>(gpl + off) = h;
                *reinterpret_cast<uint32_t*>(gpl + PLANE + off) = l;
                f16_split2(pr[p][0][j] * ps, pr[p][1][j] * ps, h, l);
                *reinterpret_cast<uint32_t*>(ppl + off) = h;
                *reinterpret_cast<uint32_t*>(ppl + PLANE + off) = l;
            }
        }
    };
    auto frag = [&](const char* plane, int col0, int s) -> f16x8 {
        const int col = col0 + i;
        return *reinterpret_cast<const f16x8*>(plane + col * 80 + (((2 * s + kg) ^ ((col >> 4) & 3)) << 4));
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int c = 0; c < nchunk; ++c) {
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunk) load_chunk(c + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 Ah[MT], Al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int col0 = wr * (H / 4) + 32 * mt;
                Ah[mt] = frag(gpl, col0, s); Al[mt] = frag(gpl + PLANE, col0, s);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col0 = wc * (H / 2) + 32 * nt;
                const f16x8 Bh = frag(ppl, col0, s), Bl = frag(ppl + PLANE, col0, s);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[mt], Bl, acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[mt], Bh, acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[mt], Bh, acc[mt][nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    float* out = a.ws + (size_t)z * H * H;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * (H / 4) + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kg;
                __builtin_nontemporal_store(acc[mt][nt][r] * unscale, out + (size_t)row * H + wc * (H / 2) + 32 * nt + i);
            }
}
