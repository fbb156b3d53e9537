// How fast can the LDS -> MFMA stream of the fp16x3 edge kernel run by itself (round 5)?  The edge kernel with everything but
// its MFMAs and W2 fragment reads ablated (HD_ABLATE=15) takes 68.8 us where the matrix pipe needs 42: this isolates the loop.
// Every wave: per "chunk" 48 v_mfma_f32_32x32x16_f16 on 8 accumulators in 8 groups of 6, each group fed by 4 ds_read_b128
// fragments (1 KiB per wave each) from a 32 KB LDS image.  WPS waves per SIMD (256-thread workgroups, WPS per CU).
//   DEPTH 0: no LDS reads (register operands) - the pipe's own rate;  1: fragments requested one group ahead (the kernel's
//   scheme);  2: two groups ahead (a third register set);  BAR: a workgroup barrier per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define DEVINL __device__ __forceinline__
template <int G> struct IC { static constexpr int value = G; };
template <int I, int N, typename F> DEVINL void static_for(F&& f) { if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); } }

template <unsigned O0, unsigned O1, unsigned O2, unsigned O3>
DEVINL void lds_read4(f16x8 (&f)[4], unsigned addr) {
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(addr), "i"(O0), "i"(O1), "i"(O2), "i"(O3));
}
template <int N> DEVINL void lds_wait4(f16x8 (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "i"(N)); }

template <int DEPTH, bool BAR, int WPS, int ILV>
__global__ __launch_bounds__(256, WPS) void k(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // 2 x 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) smem[i] = in[i & 4095] * 1e-3f;
    __syncthreads();
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f16x8 ah, al;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)in[lane + i]; al[i] = (_Float16)(in[lane + 8 + i] * 1e-3f); }
    const unsigned w_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const unsigned wb = w_lds + (it & 1) * 32768;
        f16x8 f0[4], f1[4], f2[4];
        if constexpr (DEPTH == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { f0[q] = ah; f1[q] = al; }
        }
        if constexpr (DEPTH >= 1) lds_read4<0, 1024, 2048, 3072>(f0, wb);
        if constexpr (DEPTH >= 2) lds_read4<4096, 4096 + 1024, 4096 + 2048, 4096 + 3072>(f1, wb);
        static_for<0, 8>([&](auto Gc) {
            constexpr int g = decltype(Gc)::value;
            constexpr int NS = DEPTH >= 2 ? 3 : 2;
            f16x8(&cur)[4] = (g % NS) == 0 ? f0 : (g % NS) == 1 ? f1 : f2;
            f16x8(&nxt)[4] = ((g + DEPTH) % NS) == 0 ? f0 : ((g + DEPTH) % NS) == 1 ? f1 : f2;
            if constexpr (DEPTH == 1) {
                lds_wait4<0>(cur);
                if constexpr (g + 1 < 8) lds_read4<(g + 1) * 4096, (g + 1) * 4096 + 1024, (g + 1) * 4096 + 2048, (g + 1) * 4096 + 3072>(nxt, wb);
            }
            if constexpr (DEPTH == 2) {
                if constexpr (g + 1 < 8) lds_wait4<4>(cur); else lds_wait4<0>(cur);
                if constexpr (g + 2 < 8) lds_read4<(g + 2) * 4096, (g + 2) * 4096 + 1024, (g + 2) * 4096 + 2048, (g + 2) * 4096 + 3072>(nxt, wb);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int c0 = (2 * g) & 7, c1 = (2 * g + 1) & 7;
            if constexpr (ILV == 2) {
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[2], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur[2], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[1], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[3], acc[c1], 0, 0, 0);
            } else {                // the three MFMAs of an accumulator back to back
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[0], acc[c0], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur[0], acc[c0], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[1], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[2], acc[c1], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, cur[2], acc[c1], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, cur[3], acc[c1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (BAR) asm volatile("s_barrier" ::: "memory");
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int DEPTH, bool BAR, int WPS, int ILV>
void run(const char* name, float* out, const float* in) {
    const int iters = 4000;
    const size_t lds = 65536;
    CK(hipFuncSetAttribute((const void*)k<DEPTH, BAR, WPS, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<DEPTH, BAR, WPS, ILV>), dim3(256 * WPS), dim3(256), lds, 0, out, in, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<DEPTH, BAR, WPS, ILV>), dim3(256 * WPS), dim3(256), lds, 0, out, in, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-64s %d waves/SIMD  %8.3f ms  %6.1f ns per MFMA per SIMD\n", name, WPS, ms, ms * 1e6 / ((double)iters * 48 * WPS));
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 1 << 22));
    float* h = (float*)malloc(1 << 20);
    srand(1);
    for (int i = 0; i < (1 << 18); ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice));
    run<0, false, 1, 2>("register operands, no LDS", out, in);
    run<0, false, 2, 2>("register operands, no LDS", out, in);
    run<1, false, 1, 2>("fragments one group ahead", out, in);
    run<1, false, 2, 2>("fragments one group ahead", out, in);
    run<2, false, 1, 2>("fragments two groups ahead", out, in);
    run<2, false, 2, 2>("fragments two groups ahead", out, in);
    run<1, true, 2, 2>("one group ahead + barrier per chunk", out, in);
    run<2, true, 2, 2>("two groups ahead + barrier per chunk", out, in);
    run<1, false, 2, 1>("one group ahead, an accumulator's 3 MFMAs back to back", out, in);
    run<0, false, 2, 1>("register operands, an accumulator's 3 MFMAs back to back", out, in);
    return 0;
}
