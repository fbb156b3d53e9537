// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs on gfx950?  And a GEMM-level check of the fp16 two-way split
// (hH + hL + lH) against fp64, next to bf16x3 / bf16x6, with the real matrix instruction.
// build: hipcc --offload-arch=gfx950 -O2 -o scratch/mb/f16_denorm scratch/mb/f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_denorm(float* out, float a_val, float b_val) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

// C[M][N] = P[M][K] W[N][K]^T, one wave per 32x32 tile; mode 0: bf16 hh+lh+hl, 1: bf16 six terms, 2: fp16 hH+lH+hL, 3: fp16 + lL
template <int MODE>
__global__ void k_gemm(const float* P, const float* W, float* C, int M, int N, int K, float sp, float sw) {
    const int lane = threadIdx.x & 63, hh = lane >> 5, n = lane & 31;
    const int tm = blockIdx.x, tn = blockIdx.y;
    f16v acc = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        float a[8], b[8];
        for (int i = 0; i < 8; ++i) { a[i] = P[(size_t)(32 * tm + n) * K + k0 + 8 * hh + i] * sp; b[i] = W[(size_t)(32 * tn + n) * K + k0 + 8 * hh + i] * sw; }
        if constexpr (MODE >= 2) {
            h8 ah, al, bh, bl;
            for (int i = 0; i < 8; ++i) {
                ah[i] = (_Float16)a[i]; al[i] = (_Float16)(a[i] - (float)ah[i]);
                bh[i] = (_Float16)b[i]; bl[i] = (_Float16)(b[i] - (float)bh[i]);
            }
            if (MODE == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        } else {
            b8 ap[3], bp[3];
            for (int i = 0; i < 8; ++i) {
                float ra = a[i], rb = b[i];
                for (int p = 0; p < 3; ++p) { ap[p][i] = (__bf16)ra; ra -= (float)ap[p][i]; bp[p][i] = (__bf16)rb; rb -= (float)bp[p][i]; }
            }
            if (MODE == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], acc, 0, 0, 0);
        }
    }
    const float inv = 1.0f / (sp * sw);
    for (int r = 0; r < 16; ++r) {
        const int row = (r / 4) * 8 + hh * 4 + (r % 4);
        C[(size_t)(32 * tm + row) * N + 32 * tn + n] = acc[r] * inv;
    }
}

int main() {
    float* d; hipMalloc(&d, 4);
    const float cases[4][2] = {{9.5367431640625e-07f /*2^-20*/, 1024.f}, {5.9604644775390625e-08f /*2^-24*/, 1024.f}, {1024.f, 9.5367431640625e-07f}, {6.103515625e-05f /*2^-14 normal*/, 1024.f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        float o; hipMemcpy(&o, d, 4, hipMemcpyDeviceToHost);
        printf("a = %.3e b = %.3e : mfma f16 sum over K = 16 -> %.6e (exact %.6e)\n", c[0], c[1], o, 16.0 * (double)c[0] * c[1]);
    }
    const int M = 2048, N = 256, K = 256;
    std::mt19937 g(0); std::normal_distribution<float> nd(0, 2); std::uniform_real_distribution<float> ud(-1.f / 16, 1.f / 16);
    for (float gp : {1.0f, 1e-2f, 30.f}) for (float gw : {1.0f, 1e-2f}) {
        std::vector<float> P((size_t)M * K), W((size_t)N * K), C((size_t)M * N);
        for (auto& v : P) { float x = nd(g); v = x / (1 + std::exp(-x)) * gp; }
        for (auto& v : W) v = ud(g) * gw;
        float wmax = 0; for (auto v : W) wmax = std::max(wmax, std::fabs(v));
        const float sw14 = std::ldexp(1.0f, 14 - (int)std::floor(std::log2(wmax)));
        std::vector<double> ref((size_t)M * N);
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)P[(size_t)i * K + k] * W[(size_t)j * K + k]; ref[(size_t)i * N + j] = s; }
        float *dP, *dW, *dC; hipMalloc(&dP, P.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, C.size() * 4);
        hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        auto rel = [&](const char* name) {
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double num = 0, den = 0; for (size_t i = 0; i < C.size(); ++i) { num += (C[i] - ref[i]) * (C[i] - ref[i]); den += ref[i] * ref[i]; }
            printf("   P x %-6g W x %-6g %-28s rel-L2 %.3e\n", gp, gw, name, std::sqrt(num / den));
        };
        dim3 grid(M / 32, N / 32);
        hipLaunchKernelGGL(k_gemm<0>, grid, dim3(64), 0, 0, dP, dW, dC, M, N, K, 1.f, 1.f); rel("bf16x3");
        hipLaunchKernelGGL(k_gemm<1>, grid, dim3(64), 0, 0, dP, dW, dC, M, N, K, 1.f, 1.f); rel("bf16x6");
        hipLaunchKernelGGL(k_gemm<2>, grid, dim3(64), 0, 0, dP, dW, dC, M, N, K, 1.f, 1.f); rel("fp16x3 unscaled");
        hipLaunchKernelGGL(k_gemm<2>, grid, dim3(64), 0, 0, dP, dW, dC, M, N, K, 1.f, sw14); rel("fp16x3 W@2^14");
        hipLaunchKernelGGL(k_gemm<2>, grid, dim3(64), 0, 0, dP, dW, dC, M, N, K, 16.f, sw14); rel("fp16x3 W@2^14 P x16");
        hipLaunchKernelGGL(k_gemm<3>, grid, dim3(64), 0, 0, dP, dW, dC, M, N, K, 16.f, sw14); rel("fp16x4 W@2^14 P x16");
        hipFree(dP); hipFree(dW); hipFree(dC);
    }
    return 0;
}
