// Wave specialisation on one SIMD (round 5): does a wavefront that only streams MFMAs (+ its LDS fragment reads)
// overlap with a co-resident wavefront that only produces operands on the VALU (SiLU + fp16 split + LDS writes)?
// One 512-thread workgroup per CU: waves 0-3 = "M" (matrix), waves 4-7 = "V" (vector); wave w and w + 4 share a SIMD
// (checked through HW_ID).  One iteration = one K chunk of the fp16x3 edge kernel:
//   M: 4 P-fragment reads + 40 W2-fragment reads (ds_read_b128) + 48 v_mfma_f32_32x32x16_f16 on 8 accumulators
//   V: 16 first-layer values per lane: 8 row-gather loads, pre-activation, SiLU, fp16 head/tail split, 4 ds_write_b128
// and a workgroup barrier per iteration.  MODE bit 0: M active, bit 1: V active, bit 2: M runs a 1,300-instruction VALU
// epilogue every 8th iteration (v1: the epilogue stays with the accumulators), bit 3: V also carries the epilogue work
// (v2: +162 VALU per iteration, accumulators handed over through LDS: 4 ds_write_b128 per iteration in M, 4 reads in V).
// PRIO: 0 none, 1 V at s_setprio 3, 2 M at s_setprio 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define DEVINL __device__ __forceinline__

template <unsigned O0, unsigned O1, unsigned O2, unsigned O3>
DEVINL void lds_read4(f16x8 (&f)[4], unsigned addr) {
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(addr), "i"(O0), "i"(O1), "i"(O2), "i"(O3));
}
template <int N> DEVINL void lds_wait4(f16x8 (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "i"(N)); }

DEVINL void split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
    const f16x2 hp = __builtin_convertvector((f32x2){y0, y1}, f16x2);
    const float l0 = y0 - (float)hp[0], l1 = y1 - (float)hp[1];
    hi = __builtin_bit_cast(uint32_t, hp);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){l0, l1}, f16x2));
}

template <int G> struct IC { static constexpr int value = G; };
template <int I, int N, typename F> DEVINL void static_for(F&& f) { if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); } }

template <int MODE, int PRIO>
__global__ __launch_bounds__(512, 1) void k(float* out, const float* in, int iters, int rows, unsigned* hwid) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [W2: 16384 floats = 64 KB][P ring: 4 SIMDs x 2 x 1024 floats][acc ring 4 x 1024 floats]
    __shared__ __attribute__((aligned(16))) float wrd_s[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_m = wave < 4;
    const int pair = wave & 3;
    for (int i = tid; i < 16384 + 8192 + 4096; i += 512) smem[i] = in[i & 4095] * 1e-3f;
    for (int i = tid; i < 1024; i += 512) wrd_s[i] = in[i] * 1e-3f;
    if (lane == 0) hwid[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 4);
    __syncthreads();
    const long long tc0 = __builtin_readcyclecounter();
    float* pring = smem + 16384 + pair * 2048;
    float* aring = smem + 16384 + 8192 + pair * 1024;
    float s = 0.f;
    if (is_m) {
        if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[8];
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        const unsigned w_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem + lane * 16;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE & 1) {
                const unsigned wb = w_lds + (it & 1) * 32768;
                const unsigned pb = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)(pring + (it & 1) * 1024) + lane * 16;
                f16x8 pf[4];
                lds_read4<0, 1024, 2048, 3072>(pf, pb);
                f16x8 f0[4], f1[4];
                lds_read4<0, 1024, 2048, 3072>(f0, wb);
                lds_wait4<4>(pf);
                static_for<0, 8>([&](auto Gc) {
                    constexpr int g = decltype(Gc)::value;
                    f16x8(&cur)[4] = (g & 1) ? f1 : f0;
                    f16x8(&nxt)[4] = (g & 1) ? f0 : f1;
                    lds_wait4<0>(cur);
                    if constexpr (g + 1 < 8) lds_read4<(g + 1) * 4096, (g + 1) * 4096 + 1024, (g + 1) * 4096 + 2048, (g + 1) * 4096 + 3072>(nxt, wb);
                    __builtin_amdgcn_sched_barrier(0);        // keep the MFMAs below the request (hipcc otherwise sinks it to save registers)
                    constexpr int c0 = (2 * g) & 7, c1 = (2 * g + 1) & 7, st = g >> 2;
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[2 * st], cur[0], acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[2 * st], cur[2], acc[c1], 0, 0, 0);
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[2 * st + 1], cur[0], acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[2 * st + 1], cur[2], acc[c1], 0, 0, 0);
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[2 * st], cur[1], acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[2 * st], cur[3], acc[c1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (MODE & 8) {          // hand one column tile of the accumulators over per iteration (4 KB)
                    const int ct = it & 7;
                    float* dst = aring + lane * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float t = 0.f;
#pragma unroll
                            for (int c = 0; c < 8; ++c) t = (ct == c) ? acc[c][4 * q + j] : t;      // wave-uniform select (the real kernel indexes statically)
                            v[j] = t;
                        }
                        *reinterpret_cast<f32x4*>(dst + q * 256) = v;
                    }
                }
                if constexpr (MODE & 4) {
                    if ((it & 7) == 7) {               // the epilogue of a tile on its accumulators: ~6 per value + the per-node sums
                        float dot[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) dot[r] = 0.f;
                        const float wav = wrd_s[lane], rs = wrd_s[64 + lane], b2v = wrd_s[128 + lane];
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            float e[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[c][r] = __builtin_fmaf(acc[c][r], rs, b2v);
#pragma unroll
                            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(acc[c][r]);
#pragma unroll
                            for (int r = 0; r < 16; ++r) e[r] = 1.0f + e[r];
#pragma unroll
                            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_rcpf(e[r]);
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[c][r] *= e[r];
#pragma unroll
                            for (int r = 0; r < 16; ++r) dot[r] = __builtin_fmaf(acc[c][r], wav, dot[r]);
                        }
                        float sm[3] = {0.f, 0.f, 0.f};
#pragma unroll
                        for (int sgi = 0; sgi < 3; ++sgi)
#pragma unroll
                            for (int c = 0; c < 8; ++c)
#pragma unroll
                                for (int r = 0; r < 16; ++r) sm[sgi] = __builtin_fmaf(dot[r], acc[c][r], sm[sgi]);
                        s += sm[0] + sm[1] + sm[2];
#pragma unroll
                        for (int c = 0; c < 8; ++c)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
                    }
                }
            }
            asm volatile("s_barrier" ::: "memory");
        }
        for (int c = 0; c < 8; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    } else {
        if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        const int hh = lane >> 5, n = lane & 31;
        // rows of one "molecule" per workgroup (30 rows = 60 KB: L2 / L1 hits, as in the edge kernel)
        const int ri = (blockIdx.x * 30 + (pair * 7 + n / 4) % 30) % rows, rj = (blockIdx.x * 30 + (pair * 11 + n * 3) % 30) % rows;
        const float* Arow = in + (size_t)ri * 512 + 16 * hh;
        const float* Brow = in + (size_t)rj * 512 + 256 + 16 * hh;
        const float radial = in[lane] + 1.f, d0 = in[lane + 64] + 2.f, finv = 0.5f;
        f32x4 pa[4], pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pa[u] = *reinterpret_cast<const f32x4*>(Arow + 4 * u); pb[u] = *reinterpret_cast<const f32x4*>(Brow + 4 * u); }
        float dotv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) dotv[r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE & 2) {
                const int c = it & 7;
                u32x4 ph[2], pl[2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wrd_s + 32 * c + 16 * hh + 4 * u);
                    const f32x4 wd4 = *reinterpret_cast<const f32x4*>(wrd_s + 256 + 32 * c + 16 * hh + 4 * u);
                    float pre[4], e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre[j] = pa[u][j] + pb[u][j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(radial, wr4[j], pre[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(d0, wd4[j], pre[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(pre[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = __builtin_fmaf(e[j], finv, finv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre[j] *= e[j];
                    uint32_t hi[2], lo[2];
                    split2(pre[0], pre[1], hi[0], lo[0]);
                    split2(pre[2], pre[3], hi[1], lo[1]);
                    ph[u >> 1][2 * (u & 1)] = hi[0]; ph[u >> 1][2 * (u & 1) + 1] = hi[1];
                    pl[u >> 1][2 * (u & 1)] = lo[0]; pl[u >> 1][2 * (u & 1) + 1] = lo[1];
                    // rows of the next chunk
                    const int cn = (it + 1) & 7;
                    pa[u] = *reinterpret_cast<const f32x4*>(Arow + 32 * cn + 4 * u);
                    pb[u] = *reinterpret_cast<const f32x4*>(Brow + 32 * cn + 4 * u);
                }
                float* dst = pring + ((it + 1) & 1) * 1024 + lane * 4;
                *reinterpret_cast<u32x4*>(dst) = ph[0];
                *reinterpret_cast<u32x4*>(dst + 256) = pl[0];
                *reinterpret_cast<u32x4*>(dst + 512) = ph[1];
                *reinterpret_cast<u32x4*>(dst + 768) = pl[1];
                if constexpr (MODE & 8) {          // one column tile of the epilogue: 16 values x (un-scale, SiLU, dot) + a share of the sums
                    const float* src = aring + lane * 4;
                    const float wav = wrd_s[lane], rs = wrd_s[64 + lane], b2v = wrd_s[128 + lane];
                    float a16[16], e[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(src + q * 256);
#pragma unroll
                        for (int j = 0; j < 4; ++j) a16[4 * q + j] = v[j];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) a16[r] = __builtin_fmaf(a16[r], rs, b2v);
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(a16[r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = 1.0f + e[r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_rcpf(e[r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) a16[r] *= e[r];
#pragma unroll
                    for (int r = 0; r < 16; ++r) dotv[r & 7] = __builtin_fmaf(a16[r], wav, dotv[r & 7]);
#pragma unroll
                    for (int sgi = 0; sgi < 3; ++sgi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) dotv[(r + sgi) & 7] = __builtin_fmaf(a16[r], rs, dotv[(r + sgi) & 7]);
                }
            }
            asm volatile("s_barrier" ::: "memory");
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += pa[u][0] + pb[u][1];
#pragma unroll
        for (int r = 0; r < 8; ++r) s += dotv[r];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && tid == 0) { hwid[4000] = (unsigned)((__builtin_readcyclecounter() - tc0) >> 8); }
}

template <int MODE, int PRIO>
void run(const char* name, float* out, const float* in, unsigned* hwid, int grid, double ghz) {
    const int iters = 4000, rows = 7680;
    const size_t lds = (16384 + 8192 + 4096) * 4;
    CK(hipFuncSetAttribute((const void*)k<MODE, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(grid), dim3(512), lds, 0, out, in, iters, rows, hwid);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, PRIO>), dim3(grid), dim3(512), lds, 0, out, in, iters, rows, hwid);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    unsigned cyc8; CK(hipMemcpy(&cyc8, hwid + 4000, 4, hipMemcpyDeviceToHost));
    const double ns = ms * 1e6 / iters, cyc = (double)cyc8 * 256.0 / iters;
    printf("%-58s %8.3f ms  %7.1f ns / chunk  = %6.0f shader cycles (%.2f GHz)  (48 MFMAs = 1536)\n", name, ms, ns, cyc, cyc / ns);
}

int main() {
    float *in, *out; unsigned* hwid;
    const size_t nin = (size_t)7680 * 512;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&hwid, 4096 * 4));
    float* h = (float*)malloc(nin * 4);
    srand(1);
    for (size_t i = 0; i < nin; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(in, h, nin * 4, hipMemcpyHostToDevice));
    const double ghz = 2.0;
    const int grid = 256;
    run<1, 0>("M alone (MFMAs + fragment reads + barrier)", out, in, hwid, grid, ghz);
    run<2, 0>("V alone (operand generation + barrier)", out, in, hwid, grid, ghz);
    run<3, 0>("M + V", out, in, hwid, grid, ghz);
    run<3, 1>("M + V, V at prio 3", out, in, hwid, grid, ghz);
    run<3, 2>("M + V, M at prio 3", out, in, hwid, grid, ghz);
    run<5, 0>("M alone with its epilogue every 8 chunks", out, in, hwid, grid, ghz);
    run<7, 0>("v1: M (+ epilogue) + V", out, in, hwid, grid, ghz);
    run<7, 1>("v1, V at prio 3", out, in, hwid, grid, ghz);
    run<7, 2>("v1, M at prio 3", out, in, hwid, grid, ghz);
    run<10, 0>("V alone carrying the epilogue too", out, in, hwid, grid, ghz);
    run<11, 0>("v2: M (+ accumulator hand-over) + V (+ epilogue)", out, in, hwid, grid, ghz);
    run<11, 1>("v2, V at prio 3", out, in, hwid, grid, ghz);
    run<11, 2>("v2, M at prio 3", out, in, hwid, grid, ghz);
    unsigned hw[16];
    CK(hipMemcpy(hw, hwid, 64, hipMemcpyDeviceToHost));
    printf("HW_ID simd of waves 0..7 (block 0):");
    for (int w = 0; w < 8; ++w) printf(" %u", (hw[w] >> 4) & 3);
    printf("   (block 1):");
    for (int w = 8; w < 16; ++w) printf(" %u", (hw[w] >> 4) & 3);
    printf("\n");
    return 0;
}
