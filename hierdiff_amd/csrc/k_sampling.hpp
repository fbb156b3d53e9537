// Small kernels around the EGNN body: coordinate update, fp32-mode neighbour-sum reduction, output stage, posterior
// step, final decode, noise, graph-replay step counter.  Included through kernels.hpp.
#pragma once
#include "common.hpp"

// ----------------------------------------------------------------------------- coordinate update
// x_i <- (x_i + sum_parts / normalization_factor) * mask_i   (egnn_new.py:100-110)

struct XupdArgs {
    const float* part;    // [P][4]
    const int* pstart;    // [M+1]
    const float* nmask;
    float* xcur;          // [M_pad][4]
    float norm;
    int M;
};

__global__ void k_xupd(XupdArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.M) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int p = a.pstart[i]; p < a.pstart[i + 1]; ++p) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a.part + (size_t)p * 4);
        sx += v[0]; sy += v[1]; sz += v[2];
    }
    float m = a.nmask[i];
    f32x4 x = *reinterpret_cast<const f32x4*>(a.xcur + (size_t)i * 4);
    x[0] = (x[0] + sx / a.norm) * m;
    x[1] = (x[1] + sy / a.norm) * m;
    x[2] = (x[2] + sz / a.norm) * m;
    *reinterpret_cast<f32x4*>(a.xcur + (size_t)i * 4) = x;
}

// agg_i = (sum of node i's partial neighbour sums, fixed order) / normalization_factor   (egnn_new.py:52-56,280-282)
struct AggArgs {
    const float* part;    // [P][H]
    const int* pstart;    // [M+1]
    float* agg;           // [M_pad][H]
    float norm;
    int M, H;
};

__global__ void k_agg(AggArgs a) {
    const int q = a.H >> 2;                                   // float4 per row
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = idx / q, c4 = idx - i * q;
    if (i >= a.M) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int p = a.pstart[i]; p < a.pstart[i + 1]; ++p) v += *reinterpret_cast<const f32x4*>(a.part + (size_t)p * a.H + 4 * c4);
    *reinterpret_cast<f32x4*>(a.agg + (size_t)i * a.H + 4 * c4) = v / a.norm;
}

// ----------------------------------------------------------------------------- output stage
// per node: embedding_out (only the F kept columns), vel = (x_final - x_in)*mask, NaN detection
// (egnn_new.py:202-204, en_dynamics.py:83-111).  One wavefront per node.

struct Post1Args {
    const float* h;       // [M_pad][H]
    const float* outW;    // [fin][H]
    const float* out_b;   // [fin]
    const float* x0;
    const float* xcur;
    const int* node_of;
    const float* nmask;
    float* out;           // [B*N][D]
    int* nanflag;
    int M, N, D, F, H, mol_shape;
};

// One wavefront per node.  The node's h row is read once (float4 per lane); up to 8 outputs at a time are
// reduced over the 64 lanes with a transpose-reduce (each exchange halves the values a lane still carries:
// 4 + 2 + 1 shuffles, then a 3-step butterfly) instead of 8 x 6 butterflies.
__global__ void k_post1(Post1Args a) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= a.M) return;
    const int flat = a.node_of[i];
    const float m = a.nmask[i];
    float* orow = a.out + (size_t)flat * a.D;
    const int q = a.H >> 2;
    for (int f0 = 0; f0 < a.F; f0 += 8) {
        float p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] = 0.f;
        for (int c4 = lane; c4 < q; c4 += 64) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(a.h + (size_t)i * a.H + 4 * c4);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (f0 + k < a.F) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(a.outW + (size_t)(f0 + k) * a.H + 4 * c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) p[k] = __builtin_fmaf(hv[j], w[j], p[k]);
                }
            }
        }
        const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
        float v4[4], v2[2], v1;
#pragma unroll
        for (int k = 0; k < 4; ++k) v4[k] = (b5 ? p[k + 4] : p[k]) + __shfl_xor(b5 ? p[k] : p[k + 4], 32);
#pragma unroll
        for (int k = 0; k < 2; ++k) v2[k] = (b4 ? v4[k + 2] : v4[k]) + __shfl_xor(b4 ? v4[k] : v4[k + 2], 16);
        v1 = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
        v1 += __shfl_xor(v1, 4); v1 += __shfl_xor(v1, 2); v1 += __shfl_xor(v1, 1);
        // lanes with bits (b5, b4, b3) hold output f0 + 4 b5 + 2 b4 + b3
        const int f = f0 + (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0);
        if ((lane & 7) == 0 && f < a.F) orow[3 + f] = (v1 + a.out_b[f]) * m;
    }
    if (lane < 3) {
        const int nloc = flat % a.N;
        float v = 0.f;
        if (a.mol_shape < 0 || nloc < a.mol_shape) v = (a.xcur[(size_t)i * 4 + lane] - a.x0[(size_t)i * 4 + lane]) * m;
        orow[lane] = v;
        if (v != v) atomicOr(a.nanflag, 1);
    }
}

// per molecule: NaN reset, centre-of-gravity removal over all N nodes, zero rows of inactive
// nodes (en_dynamics.py:109-116, models/utils.py:43-57).  One wavefront per molecule.

struct Post2Args {
    const int* slot_of;   // [B*N] compact id or -1
    const float* nmask;   // [M_pad]
    const int* nvalid;    // [B] count of node_mask
    float* out;           // [B*N][D]
    const int* nanflag;
    long long* nan_events;
    int B, N, D;
};

__global__ void k_post2(Post2Args a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const bool nan = (*a.nanflag) != 0;
    if (nan && b == 0 && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.nan_events), 1ULL);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int nn = lane; nn < a.N; nn += 64) {
        const int flat = b * a.N + nn;
        float* orow = a.out + (size_t)flat * a.D;
        if (a.slot_of[flat] < 0) {
            for (int d = 0; d < a.D; ++d) orow[d] = 0.f;
        } else if (nan) {
            orow[0] = 0.f; orow[1] = 0.f; orow[2] = 0.f;
        } else {
            sx += orow[0]; sy += orow[1]; sz += orow[2];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
    const int cnt = a.nvalid[b];
    if (cnt == 0) return;
    const float mx = sx / (float)cnt, my = sy / (float)cnt, mz = sz / (float)cnt;
    for (int nn = lane; nn < a.N; nn += 64) {
        const int flat = b * a.N + nn;
        const int s = a.slot_of[flat];
        if (s < 0) continue;
        const float m = a.nmask[s];
        float* orow = a.out + (size_t)flat * a.D;
        orow[0] -= mx * m; orow[1] -= my * m; orow[2] -= mz * m;
    }
}

// ----------------------------------------------------------------------------- sampling maths
// One wavefront per molecule; all reductions are over <= N nodes.

struct NoiseSrc {
    const float* raw_x;   // [rows][mol][3] or null -> Philox
    const float* raw_h;   // [rows][mol][F]
    int rows;             // 1 = shared row (fix_noise)
    uint64_t seed, sample_base;
    uint32_t draw;
    int share;            // Philox: all rows use sample_base
};

HD_DEVINL float raw_noise(const NoiseSrc& s, int b, int nn, int c, int mol, int F) {
    if (s.raw_x) {
        const int rb = (s.rows == 1) ? 0 : b;
        return (c < 3) ? s.raw_x[((size_t)rb * mol + nn) * 3 + c] : s.raw_h[((size_t)rb * mol + nn) * F + (c - 3)];
    }
    const uint64_t sid = s.sample_base + (s.share ? 0 : (uint64_t)b);
    return philox_normal(s.seed, sid, s.draw, (uint32_t)(nn * (3 + F) + c));
}

struct StepArgs {
    const float* zt;      // [B][N][D]
    const float* eps;     // [B][N][D]
    const float* coef;    // [rows][4]
    const uint8_t* nm;    // [B*N] node mask bytes
    float* zs;            // [B][out_stride][D]
    NoiseSrc noise;
    const uint32_t* draw_ptr;   // optional device-side draw counter (graph replay); overrides noise.draw
    const int* step_ptr;        // optional device-side step index into coef (graph replay)
    const unsigned long long* base_ptr;   // optional device-side first global sample id (graph replay); overrides noise.sample_base
    uint32_t draw0;             // draw index of the first replayed step (raw-noise offset base)
    int coef_rows, B, N, D, F, mol, out_stride;
};

// sum of `v` over the workgroup (4 wavefronts), result in every thread; `red` is 4 floats of LDS scratch
HD_DEVINL float block_sum4(float v, float* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                               // previous use of `red` is over
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// sample_p_zs_given_zt after the network call (diffusion_qm9.py:326-345) + sample_normal.
// One workgroup (256 threads) per molecule, one thread per (node, component): every raw normal (Philox + Box-Muller
// is ~150 instructions) is produced once and kept in LDS for the second pass; dynamic LDS = mol * D floats.
__global__ __launch_bounds__(256) void k_post_step(StepArgs a) {
    extern __shared__ float nz_s[];                // [mol * D] masked raw normals, later the un-centred z_s
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    NoiseSrc ns = a.noise;
    if (a.base_ptr) ns.sample_base = *a.base_ptr;
    if (a.draw_ptr) {
        ns.draw = *a.draw_ptr;
        if (ns.raw_x) {
            const size_t k = (size_t)(ns.draw - a.draw0) * ns.rows * a.mol;
            ns.raw_x += k * 3;
            ns.raw_h += k * a.F;
        }
    }
    const float* cf = a.coef + (a.step_ptr ? (size_t)(*a.step_ptr) * 4 : (size_t)((a.coef_rows == 1) ? 0 : b) * 4);
    const float alpha_ts = cf[0], sigma2_ts = cf[1], sigma_t = cf[2], sigma = cf[3];
    const float ceps = (sigma2_ts / alpha_ts) / sigma_t;
    const int mol = a.mol, D = a.D, total = mol * D;
    // pass 1: raw normals (masked) into LDS; masked sums of eps_x and of the x-noise per component, node count
    float se[3] = {0.f, 0.f, 0.f}, sn[3] = {0.f, 0.f, 0.f}, cnt = 0.f;
    for (int e = tid; e < total; e += 256) {
        const int nn = e / D, c = e - nn * D;
        const float m = a.nm[b * a.N + nn] ? 1.f : 0.f;
        const float z = raw_noise(ns, b, nn, c, mol, a.F) * m;
        nz_s[e] = z;
        if (c < 3) {
            const float ev = a.eps[((size_t)b * a.N + nn) * D + c];
#pragma unroll
            for (int k = 0; k < 3; ++k) { if (c == k) { se[k] += ev; sn[k] += z; } }
            if (c == 0) cnt += m;
        }
    }
    float em[3], nmn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { em[k] = block_sum4(se[k], red, tid); nmn[k] = block_sum4(sn[k], red, tid); }
    cnt = block_sum4(cnt, red, tid);
#pragma unroll
    for (int k = 0; k < 3; ++k) { em[k] /= cnt; nmn[k] /= cnt; }
    // pass 2: z_s before the final re-centring (kept in LDS); its x sums
    float sv[3] = {0.f, 0.f, 0.f};
    for (int e = tid; e < total; e += 256) {
        const int nn = e / D, c = e - nn * D;
        const float m = a.nm[b * a.N + nn] ? 1.f : 0.f;
        const float zt = a.zt[((size_t)b * a.N + nn) * D + c];
        float ev = a.eps[((size_t)b * a.N + nn) * D + c];
        float z = nz_s[e];
        if (c < 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { if (c == k) { ev -= em[k] * m; z -= nmn[k] * m; } }
        }
        const float v = (zt / alpha_ts - ceps * ev) + sigma * z;
        nz_s[e] = v;
        if (c < 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { if (c == k) sv[k] += v; }
        }
    }
    float mean[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) mean[k] = block_sum4(sv[k], red, tid) / cnt;
    for (int e = tid; e < total; e += 256) {
        const int nn = e / D, c = e - nn * D;
        float v = nz_s[e];
        if (c < 3) {
            const float m = a.nm[b * a.N + nn] ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { if (c == k) v -= mean[k] * m; }
        }
        a.zs[((size_t)b * a.out_stride + nn) * D + c] = v;
    }
}

// sample_p_xh_given_z0 after the network call + unnormalize with unit norm values
// (diffusion_qm9.py:302-310,174-179): x = (1/alpha_0 * (z0 - sigma_0*eps) + sigma_x*noise)[:3],
// h = z0[3:] * mask.
struct DecodeArgs {
    const float* z0;
    const float* eps;
    const uint8_t* nm;
    float* x;             // [B][N][3]
    float* hfeat;         // [B][N][F]
    NoiseSrc noise;
    float sigma_0, alpha_0, sigma_x;
    int B, N, D, F;
};

__global__ void k_final_decode(DecodeArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= a.B) return;
    float nx = 0.f, ny = 0.f, nz = 0.f, cnt = 0.f;
    for (int nn = lane; nn < a.N; nn += 64) {
        const float m = a.nm[b * a.N + nn] ? 1.f : 0.f;
        nx += raw_noise(a.noise, b, nn, 0, a.N, a.F) * m;
        ny += raw_noise(a.noise, b, nn, 1, a.N, a.F) * m;
        nz += raw_noise(a.noise, b, nn, 2, a.N, a.F) * m;
        cnt += m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nx += __shfl_xor(nx, o); ny += __shfl_xor(ny, o); nz += __shfl_xor(nz, o); cnt += __shfl_xor(cnt, o);
    }
    const float nmn[3] = {nx / cnt, ny / cnt, nz / cnt};
    const float inv_a = 1.0f / a.alpha_0;
    for (int nn = lane; nn < a.N; nn += 64) {
        const float m = a.nm[b * a.N + nn] ? 1.f : 0.f;
        const size_t r = (size_t)b * a.N + nn;
        const float* zr = a.z0 + r * a.D;
        const float* er = a.eps + r * a.D;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float nz_ = raw_noise(a.noise, b, nn, c, a.N, a.F) * m - nmn[c] * m;
            a.x[r * 3 + c] = inv_a * (zr[c] - a.sigma_0 * er[c]) + a.sigma_x * nz_;
        }
        for (int f = 0; f < a.F; ++f) a.hfeat[r * a.F + f] = zr[3 + f] * m;
    }
}

// sample_combined_position_feature_noise (diffusion_qm9.py:445-456).
struct NoiseArgs {
    const uint8_t* nm;
    float* z;             // [B][N][D]
    NoiseSrc noise;
    int B, N, D, F;
};

__global__ void k_noise(NoiseArgs a) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= a.B) return;
    float nx = 0.f, ny = 0.f, nz = 0.f, cnt = 0.f;
    for (int nn = lane; nn < a.N; nn += 64) {
        const float m = a.nm[b * a.N + nn] ? 1.f : 0.f;
        nx += raw_noise(a.noise, b, nn, 0, a.N, a.F) * m;
        ny += raw_noise(a.noise, b, nn, 1, a.N, a.F) * m;
        nz += raw_noise(a.noise, b, nn, 2, a.N, a.F) * m;
        cnt += m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nx += __shfl_xor(nx, o); ny += __shfl_xor(ny, o); nz += __shfl_xor(nz, o); cnt += __shfl_xor(cnt, o);
    }
    const float nmn[3] = {nx / cnt, ny / cnt, nz / cnt};
    for (int nn = lane; nn < a.N; nn += 64) {
        const float m = a.nm[b * a.N + nn] ? 1.f : 0.f;
        float* o = a.z + ((size_t)b * a.N + nn) * a.D;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = raw_noise(a.noise, b, nn, c, a.N, a.F) * m - nmn[c] * m;
        for (int c = 3; c < a.D; ++c) o[c] = raw_noise(a.noise, b, nn, c, a.N, a.F) * m;
    }
}

// graph-replay helpers: k_loop_state initialises the device-side step / draw / time / sample-base words (values ride
// in the kernel arguments, so no host buffer has to outlive the call), k_advance moves them on after each captured step
__global__ void k_loop_state(int* step, uint32_t* draw, float* t_cur, unsigned long long* base, const float* tau,
                             int s0, uint32_t d0, unsigned long long b0) {
    *step = s0; *draw = d0; *t_cur = tau[s0 + 1]; *base = b0;
}

__global__ void k_advance(int* step, uint32_t* draw, float* t_cur, const float* tau) {
    int s = *step - 1;
    *step = s;
    *draw = *draw + 1;
    *t_cur = tau[s + 1 >= 0 ? s + 1 : 0];
}


// ----------------------------------------------------------------------------- measurement aid: the matrix instruction's own rate
// hd_mfma_probe (round 5): every SIMD of the chip streams one MFMA opcode from REGISTER operands on eight accumulators (no LDS, no
// memory) - the rate the chip sustains for that instruction under its power budget, on the caller's data.  The edge kernels'
// matrix time is priced against it: 19.7 ns per v_mfma_f32_32x32x16_f16 and SIMD at full occupancy, not the
// 13.3 ns of 32 cycles at 2.4 GHz.  KIND 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_f16, 2: v_mfma_f32_32x32x16_bf16.
template <int KIND>
__global__ __launch_bounds__(256, 2) void k_mfma_probe(const float* in, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f16x8 ah, bh;
    bf16x8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)in[(lane * 8 + i) & 1023]; bh[i] = (_Float16)in[(lane * 8 + i + 512) & 1023];
        ab[i] = (__bf16)in[(lane * 8 + i) & 1023]; bb[i] = (__bf16)in[(lane * 8 + i + 512) & 1023];
    }
    const float af = in[lane & 1023], bfv = in[(lane + 64) & 1023];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (KIND == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(af), "v"(bfv));
            if constexpr (KIND == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(ah), "v"(bh));
            if constexpr (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(ab), "v"(bb));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
