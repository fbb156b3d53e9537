#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void copy4(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read4(const f32x4* __restrict__ a, float* out, size_t n, int reps) {
    f32x4 s = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s[0] == 1.2345f) out[0] = s[1];
}
__global__ void empty(float* out) { if (out == nullptr) out[0] = 1; }
int main() {
    f32x4 *a, *b; float* o;
    size_t maxb = 256u << 20;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto fn, int n) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < n; ++i) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / n * 1e3; };
    printf("empty kernel: %.2f us per launch (back-to-back)\n", timeit([&] { hipLaunchKernelGGL(empty, dim3(256), dim3(256), 0, 0, o); }, 200));
    for (size_t mb : {1, 2, 4, 8, 16, 32, 64, 128}) {
        size_t n = (mb << 20) / 16;
        for (int grid : {512, 2048}) {
            float us = timeit([&] { hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, a, b, n); }, 50);
            float ur = timeit([&] { hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, a, o, n, 1); }, 50);
            float ur8 = timeit([&] { hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, a, o, n, 8); }, 20);
            printf("%4zu MB grid %4d: copy %.1f us (%.2f TB/s r+w)   read %.1f us (%.2f TB/s)   read x8 %.1f us (%.2f TB/s)\n", mb, grid, us, 2.0 * mb * 1.048576 / us, ur, mb * 1.048576 / ur, ur8, 8.0 * mb * 1.048576 / ur8);
        }
    }
    return 0;
}
