// Micro-benchmarks for the edge-kernel inner loop on gfx950: which ingredient costs what.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: pure MFMA, 8 independent accumulators, operands in registers
// MODE 1: + 3-long dependent chains (same acc thrice)
// MODE 2: MODE 1 + B fragments read from LDS (compiler-placed)
// MODE 3: MODE 1 + asm-pinned double-buffered LDS reads
// MODE 4: MODE 3 + ~180 VALU per 48 MFMA (SiLU-like)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) smem[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a0, a1, b0;
    for (int i = 0; i < 8; ++i) { a0[i] = (__bf16)in[lane + i]; a1[i] = (__bf16)in[lane + 8 + i]; b0[i] = (__bf16)in[lane + 16 + i]; }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[lane + i];
    const __bf16* wb = reinterpret_cast<const __bf16*>(smem);
    (void)wb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* big = in;                        // "AB" rows: 16 MB region
    const int rowA = (blockIdx.x * 7 + (lane & 31) * 0) & 8191, rowB = (blockIdx.x * 29 + (lane & 31)) & 8191;
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ pa[4], pb[4];
    for (int u = 0; u < 4; ++u) { pa[u] = f32x4_{0, 0, 0, 0}; pb[u] = pa[u]; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE >= 5) {
            __syncthreads();
            const float* src = in + ((it + 1) & 7) * 8192;
            float* dst = smem + ((it + 1) & 1) * 8192;
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int piece = wave * 8 + u; glds16(src + piece * 256 + lane * 4, dst + piece * 256); }
            wb = reinterpret_cast<const __bf16*>(smem + (it & 1) * 8192);
        }
        if constexpr (MODE >= 6) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += pa[i >> 2][i & 3] + pb[i >> 2][i & 3];
            const int c = (it + 2) & 7;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pa[u] = *reinterpret_cast<const f32x4_*>(big + (size_t)rowA * 512 + 32 * c + 16 * (lane >> 5) + 4 * u);
                pb[u] = *reinterpret_cast<const f32x4_*>(big + (size_t)rowB * 512 + 256 + 32 * c + 16 * (lane >> 5) + 4 * u);
            }
        }
        if constexpr (MODE == 0) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[c], 0, 0, 0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[c], 0, 0, 0);
                }
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    bf16x8 bh = *reinterpret_cast<const bf16x8*>(wb + ((rep * 8 + c) * 64 + lane) * 8);
                    bf16x8 bl = *reinterpret_cast<const bf16x8*>(wb + (((2 + rep) * 8 + c) * 64 + lane) * 8);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bh, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bh, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bl, acc[c], 0, 0, 0);
                }
        } else {
            const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)wb + lane * 16;
            bf16x8 f0[4], f1[4];
            auto rd = [&](bf16x8 (&f)[4], int g) {
                unsigned ad = base + g * 4096;
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                             : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(ad));
            };
            rd(f0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                bf16x8(&cur)[4] = (g & 1) ? f1 : f0;
                bf16x8(&nxt)[4] = (g & 1) ? f0 : f1;
                if (g < 7) { rd(nxt, g + 1); asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])); }
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
                const int c0 = (2 * g) & 7, c1 = (2 * g + 1) & 7;
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, cur[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, cur[2], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, cur[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, cur[2], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, cur[1], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, cur[3], acc[c1], 0, 0, 0);
            }
            if constexpr (MODE >= 4) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float x = v[i] + 0.001f * (float)it;
                    float e = __builtin_amdgcn_exp2f(x * -1.44269504f);
                    float y = x * __builtin_amdgcn_rcpf(1.0f + e);
                    __bf16 h = (__bf16)y;
                    float l = y - (float)h;
                    v[i] = v[i] * 0.999f + l;
                    if (i < 8) a0[i] = h; else a1[i - 8] = h;
                }
            }
        }
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int WPS>
void run(const char* name, float* out, const float* in, int grid, int iters) {
    CK(hipFuncSetAttribute((const void*)k<MODE, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 65536, 0, out, in, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 65536, 0, out, in, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double mfma = (double)grid * 4 * iters * 48;
    double tflops = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD assuming 2.4 GHz and grid spread evenly over 1024 SIMDs
    double cyc = ms * 1e-3 * 2.4e9 / (mfma / 1024.0);
    printf("%-44s grid %4d  %.3f ms  %.0f TFLOP/s  %.1f cyc/MFMA/SIMD@2.4GHz\n", name, grid, ms, tflops, cyc);
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 32 << 20)); CK(hipMalloc(&out, 1 << 22));
    std::vector<float> h(8 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    CK(hipMemcpy(in, h.data(), 32 << 20, hipMemcpyHostToDevice));
    const int iters = 400;
    run<0, 2>("0 pure MFMA, 2 waves/SIMD", out, in, 512, iters);
    run<0, 1>("0 pure MFMA, 1 wave/SIMD", out, in, 256, iters);
    run<1, 2>("1 3-chains, 2 w/SIMD", out, in, 512, iters);
    run<1, 1>("1 3-chains, 1 w/SIMD", out, in, 256, iters);
    run<2, 2>("2 + LDS frags (compiler), 2 w/SIMD", out, in, 512, iters);
    run<3, 2>("3 + LDS frags (asm pipelined), 2 w/SIMD", out, in, 512, iters);
    run<3, 1>("3 + LDS frags (asm pipelined), 1 w/SIMD", out, in, 256, iters);
    run<4, 2>("4 + VALU silu/split, 2 w/SIMD", out, in, 512, iters);
    run<4, 1>("4 + VALU silu/split, 1 w/SIMD", out, in, 256, iters);
    run<5, 2>("5 + per-chunk barrier + glds stream, 2 w/SIMD", out, in, 512, iters);
    run<6, 2>("6 + row gathers 2 chunks ahead, 2 w/SIMD", out, in, 512, iters);
    run<5, 2>("5 (grid 1740: 3.4 rounds of short WGs, iters=8)", out, in, 1740, 8);
    run<6, 2>("6 (grid 1740, iters=8)", out, in, 1740, 8);
    run<5, 2>("5 (grid 870, iters=16)", out, in, 870, 16);
    run<5, 2>("5 (grid 512, iters=27)", out, in, 512, 27);
    run<5, 2>("5 (grid 512, iters=54)", out, in, 512, 54);
    run<4, 2>("4 (grid 1740, iters=8)", out, in, 1740, 8);
    run<3, 2>("3 (grid 1740, iters=8)", out, in, 1740, 8);
    run<0, 2>("0 (grid 1740, iters=8)", out, in, 1740, 8);
    return 0;
}
