// How does VALU/MFMA overlap change with the number of wavefronts per SIMD?  Every wave runs the same stream:
// 4 independent accumulators, after each v_mfma_f32_32x32x16_bf16 NV plain VALU instructions (v_fma_f32 or a
// SiLU-like exp/add/rcp/mul mix) on unrelated registers.  grid = 256 * waves-per-SIMD workgroups of 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int NV, int MIX, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, const float* in, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a0, b0;
    for (int i = 0; i < 8; ++i) { a0[i] = (__bf16)in[lane + i]; b0[i] = (__bf16)in[lane + 16 + i]; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[lane + i] * 0.5f + 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a0), "v"(b0));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if constexpr (MIX == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
                else {
                    if ((i & 3) == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(v[i & 7]) : "v"(v[(i + 3) & 7]));
                    if ((i & 3) == 1) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[(i + 7) & 7]));
                    if ((i & 3) == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[(i + 6) & 7]));
                    if ((i & 3) == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[(i + 5) & 7]) : "v"(v[(i + 1) & 7]));
                }
            }
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int MIX, int WPS>
void run(float* out, const float* in) {
    const int iters = 4000, grid = 256 * WPS;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<NV, MIX, WPS>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<NV, MIX, WPS>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // SIMD-level time per MFMA (all waves of a SIMD together issue WPS * iters * 4 MFMAs)
    printf("  %2d %s per MFMA, %d waves/SIMD: %7.3f ms  -> %5.1f ns per MFMA per SIMD\n", NV, MIX ? "SiLU-mix VALU" : "v_fma_f32    ",
           WPS, ms, ms * 1e6 / ((double)WPS * iters * 4));
}

template <int WPS>
void sweep(float* out, const float* in) {
    run<0, 0, WPS>(out, in); run<4, 0, WPS>(out, in); run<8, 0, WPS>(out, in); run<12, 0, WPS>(out, in);
    run<4, 1, WPS>(out, in); run<8, 1, WPS>(out, in); run<12, 1, WPS>(out, in);
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(in, 0, 1 << 20));
    sweep<1>(out, in); sweep<2>(out, in); sweep<3>(out, in); sweep<4>(out, in);
    return 0;
}
