// Which VALU instructions issue in the shadow of an MFMA on gfx950?  Each loop iteration issues 8 independent
// v_mfma_f32_32x32x16_bf16 and, after every MFMA, NV VALU instructions of one kind on unrelated registers.
// If they co-issue the time per MFMA stays at the MFMA-only figure until the VALU work exceeds the MFMA's passes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int KIND, int NV>
__device__ __forceinline__ void valu(float (&v)[8], f32x2 (&p)[4]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
        if constexpr (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));
        if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
        if constexpr (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
        if constexpr (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));
        if constexpr (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i & 7]));
        if constexpr (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
        if constexpr (KIND == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
    }
}

template <int KIND, int NV, bool MF, bool AG = false>
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a0, b0;
    for (int i = 0; i < 8; ++i) { a0[i] = (__bf16)in[lane + i]; b0[i] = (__bf16)in[lane + 16 + i]; }
    float v[8]; f32x2 p[4];
    for (int i = 0; i < 8; ++i) v[i] = in[lane + i] * 0.5f;
    for (int i = 0; i < 4; ++i) p[i] = f32x2{in[lane + i], in[lane + 4 + i]} * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (MF && !AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a0), "v"(b0));
            if constexpr (MF && AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a0), "v"(b0));
            valu<KIND, NV>(v, p);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NV, bool MF, bool AG = false>
void run(const char* name, float* out, const float* in, int grid) {
    const int iters = 4000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<KIND, NV, MF, AG>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<KIND, NV, MF, AG>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // ns per (MFMA + NV VALU) slot per wave
    const double slots = (double)iters * 8;
    printf("%-34s grid %3d  %8.3f ms   %6.1f ns/slot\n", name, grid, ms, ms * 1e6 / slots);
}

#define ROW(K, NAME) \
    run<K, 4, true>(NAME " x4 + MFMA", out, in, g); run<K, 8, true>(NAME " x8 + MFMA", out, in, g); \
    run<K, 4, false>(NAME " x4 alone", out, in, g); run<K, 8, false>(NAME " x8 alone", out, in, g);

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(in, 0, 1 << 20));
    for (int g : {256, 512}) {
        printf("---- grid %d (%d wave(s) per SIMD)\n", g, g / 256);
        run<0, 0, true>("MFMA only", out, in, g);
        ROW(1, "v_fma_f32")
        run<0, 0, true, true>("MFMA only, acc in AGPRs", out, in, g);
        run<1, 4, true, true>("v_fma_f32 x4 + MFMA(AGPR)", out, in, g); run<1, 8, true, true>("v_fma_f32 x8 + MFMA(AGPR)", out, in, g);
        run<1, 12, true, false>("v_fma_f32 x12 + MFMA", out, in, g); run<1, 12, true, true>("v_fma_f32 x12 + MFMA(AGPR)", out, in, g);
        run<3, 8, true, true>("v_exp_f32 x8 + MFMA(AGPR)", out, in, g);
        ROW(2, "v_pk_fma_f32")
        ROW(5, "v_pk_mul_f32")
        ROW(3, "v_exp_f32")
        ROW(6, "v_rcp_f32")
        ROW(4, "v_cvt_pk_bf16_f32")
        ROW(7, "v_cndmask_b32")
        ROW(8, "v_and_b32")
    }
    return 0;
}
