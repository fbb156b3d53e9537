#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
// MODE 0: pure fp32 MFMA (8 accs, register operands); MODE 1: + LDS b128 fragment per 4 MFMAs (compiler placed)
// MODE 2: + 16 SiLU-like VALU per 128 MFMA
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, const float* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) smem[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[8];
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a[16], b[4];
    for (int i = 0; i < 16; ++i) a[i] = in[lane + i];
    for (int i = 0; i < 4; ++i) b[i] = in[lane + 32 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 bv[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if constexpr (MODE == 0) bv[c] = f32x4{b[0], b[1], b[2], b[3]};
                else bv[c] = *reinterpret_cast<const f32x4*>(smem + ((q * 8 + c) * 64 + lane) * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + j], bv[c][j], acc[c], 0, 0, 0);
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float x = a[i] * 0.999f + 0.001f;
                float e = __builtin_amdgcn_exp2f(x * -1.44269504f);
                a[i] = x * __builtin_amdgcn_rcpf(1.0f + e) + 0.3f;
            }
        }
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + a[3];
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
    double mfma = (double)grid * 4 * iters * 128;
    printf("%-44s grid %4d  %.3f ms  %.1f TFLOP/s\n", name, grid, ms, mfma * 2.0 * 32 * 32 * 2 / (ms * 1e-3) / 1e12);
}
int main() {
    float *in, *out;
    CK(hipMalloc(&in, 1 << 20)); CK(hipMalloc(&out, 1 << 22));
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    CK(hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice));
    const int iters = 100;
    run<0, 2>("fp32 pure MFMA, 2 w/SIMD", out, in, 512, iters);
    run<0, 1>("fp32 pure MFMA, 1 w/SIMD", out, in, 256, iters);
    run<1, 2>("fp32 + LDS frags, 2 w/SIMD", out, in, 512, iters);
    run<1, 1>("fp32 + LDS frags, 1 w/SIMD", out, in, 256, iters);
    run<2, 2>("fp32 + LDS + VALU, 2 w/SIMD", out, in, 512, iters);
    return 0;
}
