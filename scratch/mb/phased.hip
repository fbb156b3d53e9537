// Do VALU work of one wavefront and MFMA work of ANOTHER wavefront on the same SIMD overlap on gfx950?
// grid 512 x 256 threads: blocks b and b+256 share a CU (one wave of each per SIMD).  Blocks < 256 run a pure
// MFMA stream, blocks >= 256 a pure VALU stream (dependent SiLU-like chains or independent FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int VK, int PRIO>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int mf_iters, int va_iters) {
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if (blockIdx.x < 256) {
        if (mf_iters == 0) return;
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[8];
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        bf16x8 a0, b0;
        for (int i = 0; i < 8; ++i) { a0[i] = (__bf16)in[lane + i]; b0[i] = (__bf16)in[lane + 16 + i]; }
        for (int it = 0; it < mf_iters; ++it) {
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a0), "v"(b0));
        }
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        if (va_iters == 0) return;
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = in[lane + i] * 0.5f + 0.1f;
        for (int it = 0; it < va_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (VK == 0) {          // independent plain FMAs (4 per value)
                    asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1"
                                 : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                } else {                          // SiLU-like: exp, add, rcp, mul (2 transcendental of 4)
                    asm volatile("v_exp_f32 %0, %1\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0\n\tv_mul_f32 %0, %0, %1"
                                 : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                }
            }
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VK, int PRIO>
float run(float* out, const float* in, int mf, int va) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<VK, PRIO>), dim3(512), dim3(256), 0, 0, out, in, mf, va);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<VK, PRIO>), dim3(512), dim3(256), 0, 0, out, in, mf, va);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(in, 0, 1 << 20));
    const int mf = 4000;                 // 32,000 MFMAs per wave
    for (int va : {8000, 16000}) {       // x 64 VALU instructions per iteration
        float m = run<0, 0>(out, in, mf, 0), v0 = run<0, 0>(out, in, 0, va), b0 = run<0, 0>(out, in, mf, va);
        float v1 = run<1, 0>(out, in, 0, va), b1 = run<1, 0>(out, in, mf, va);
        printf("MFMA wave alone %.3f ms | FMA wave alone %.3f ms, both %.3f ms | SiLU wave alone %.3f ms, both %.3f ms\n", m, v0, b0, v1, b1);
        printf("   VALU wave at prio 3: FMA both %.3f ms, SiLU both %.3f ms;  MFMA wave at prio 3: FMA both %.3f ms\n",
               run<0, 1>(out, in, mf, va), run<1, 1>(out, in, mf, va), run<0, 2>(out, in, mf, va));
    }
    return 0;
}
