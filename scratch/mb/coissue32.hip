// How much VALU work rides in the shadow of a 16-pass fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles) inside ONE wavefront?
// Each loop iteration issues 8 independent MFMAs and, after every MFMA, NV VALU instructions on unrelated registers:
// KIND 1 = v_fma_f32, 3 = v_exp_f32, 9 = a SiLU-like mix (mul, fma, exp, fma, add, rcp, mul ...).
// 1 or 2 wavefronts per SIMD, accumulators in VGPRs or AGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int KIND, int NV>
__device__ __forceinline__ void valu(float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
        if constexpr (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
        if constexpr (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
        // KIND 11: packed fp32 (two lanes' worth of FMAs per instruction: NV packed = 2 NV scalar FMAs), on register pairs
        if constexpr (KIND == 11) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2& p = *reinterpret_cast<f32x2*>(&v[(2 * i) & 6]);
            const f32x2 q = *reinterpret_cast<const f32x2*>(&v[(2 * i + 2) & 6]);
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
        }
        if constexpr (KIND == 9) {
            if ((i % 6) == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
            else if ((i % 6) == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i & 7]));
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
        }
    }
}

template <int KIND, int NV, bool MF, bool AG>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a0 = in[lane], b0 = in[lane + 16];
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[lane + i] * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (MF && !AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a0), "v"(b0));
            if constexpr (MF && AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a0), "v"(b0));
            valu<KIND, NV>(v);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NV, bool MF, bool AG>
void run(const char* name, float* out, const float* in, int grid) {
    const int iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<KIND, NV, MF, AG>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<KIND, NV, MF, AG>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double slots = (double)iters * 8 * (grid / 256);       // MFMA slots per SIMD
    printf("%-12s NV %2d %s %s grid %3d  %8.3f ms   %6.1f ns per MFMA slot per SIMD\n", name, NV, MF ? "MFMA" : "----", AG ? "agpr" : "vgpr", grid, ms, ms * 1e6 / slots);
}

template <int KIND, bool AG>
void sweep(const char* name, float* out, const float* in, int g) {
    run<KIND, 2, true, AG>(name, out, in, g); run<KIND, 4, true, AG>(name, out, in, g); run<KIND, 6, true, AG>(name, out, in, g);
    run<KIND, 8, true, AG>(name, out, in, g); run<KIND, 12, true, AG>(name, out, in, g); run<KIND, 16, true, AG>(name, out, in, g);
    run<KIND, 8, false, AG>(name, out, in, g); run<KIND, 16, false, AG>(name, out, in, g);
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(in, 0, 1 << 20));
    for (int g : {256, 512}) {
        printf("---- grid %d (%d wave(s) per SIMD)\n", g, g / 256);
        run<0, 0, true, false>("MFMA only", out, in, g);
        run<0, 0, true, true>("MFMA only", out, in, g);
        sweep<1, false>("v_fma_f32", out, in, g);
        sweep<1, true>("v_fma_f32", out, in, g);
        sweep<9, true>("silu mix", out, in, g);
        sweep<3, true>("v_exp_f32", out, in, g);
        sweep<7, true>("v_cndmask", out, in, g);
        sweep<11, true>("v_pk_fma_f32", out, in, g);     // compare NV packed with 2 NV of v_fma_f32
    }
    return 0;
}
