// Is a K-long contraction on v_mfma_f32_16x16x4_f32 (16 x 16 tile, four k per instruction) bit-identical to the same
// contraction on v_mfma_f32_32x32x2_f32 (32 x 32 tile, two k per instruction) and to a scalar fmaf chain, when the k values are
// fed in the same order?  If so, the fp32 node GEMMs of very small batches could be spread over four times the wavefronts
// (a 32 x 32 tile's K = 512 chain is 7.4 us on one wavefront) without giving up batch-size-independent bits.
// k order of the production kernels inside a 32-wide chunk: (4q+j, 16+4q+j) for q = 0..3, j = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int K = 512;

// A [32][K], B [K][32] row-major; C32, C16, Cs [32][32]
__global__ void k32(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c = 0; c < K / 32; ++c)
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 4; ++j) {
                const int k = 32 * c + 16 * hh + 4 * q + j;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[n * K + k], B[k * 32 + n], acc, 0, 0, 0);
            }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + n] = acc[r];
}
// one wavefront per 16 x 16 quadrant (blockIdx.x = 2 * row quadrant + column quadrant)
__global__ void k16(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;        // g: k slot of the instruction
    const int r0 = 16 * (blockIdx.x >> 1), c0 = 16 * (blockIdx.x & 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < K / 32; ++c)
        for (int p = 0; p < 8; ++p) {                 // pair index p: values (2p, 2p+1) of the chunk's (4q+j) sequence
            const int v = 2 * p + (g >> 1);           // slots 0,1 -> value 2p (k, k+16); slots 2,3 -> value 2p+1
            const int k = 32 * c + v + 16 * (g & 1);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(r0 + m) * K + k], B[k * 32 + c0 + m], acc, 0, 0, 0);
        }
    for (int i = 0; i < 4; ++i) C[(r0 + 4 * g + i) * 32 + c0 + m] = acc[i];
}
__global__ void ks(const float* A, const float* B, float* C) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 1024) return;
    const int r = t >> 5, n = t & 31;
    float acc = 0.f;
    for (int c = 0; c < K / 32; ++c)
        for (int v = 0; v < 16; ++v)
            for (int h = 0; h < 2; ++h) { const int k = 32 * c + v + 16 * h; acc = __builtin_fmaf(A[r * K + k], B[k * 32 + n], acc); }
    C[t] = acc;
}
int main() {
    std::vector<float> A(32 * K), B(K * 32);
    srand(7);
    for (auto& x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto& x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 3 * 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipLaunchKernelGGL(k16, dim3(4), dim3(64), 0, 0, dA, dB, dC + 1024);
    hipLaunchKernelGGL(ks, dim3(4), dim3(256), 0, 0, dA, dB, dC + 2048);
    CK(hipDeviceSynchronize());
    std::vector<float> C(3 * 1024);
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int d16 = 0, ds = 0, d16s = 0;
    for (int i = 0; i < 1024; ++i) {
        d16 += memcmp(&C[i], &C[1024 + i], 4) != 0;
        ds += memcmp(&C[i], &C[2048 + i], 4) != 0;
        d16s += memcmp(&C[1024 + i], &C[2048 + i], 4) != 0;
    }
    printf("K = %d: elements differing  32x32x2 vs 16x16x4: %d / 1024;  32x32x2 vs fmaf chain: %d;  16x16x4 vs fmaf chain: %d\n", K, d16, ds, d16s);
    return 0;
}
