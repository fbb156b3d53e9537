// Would the fp16x3 edge kernel gain from ONE 8-wave workgroup per CU that keeps the W2 head pieces resident in LDS (128 KB) and
// streams only the tail pieces, all eight wavefronts in lockstep per chunk and persistent over tiles - instead of two independent
// 4-wave workgroups per CU that each stream head + tail (round 5: the stream costs 16 % of the kernel, profiles/r05_ablate_fp16x3.log)?
// Every wavefront runs the kernel's phases with its real instruction mix, per tile:
//   prologue  a dependent chain of three L2 loads (metadata -> coordinates -> rows) + ~300 VALU (first operands, accumulator setup)
//   loop      8 chunks x [barrier; LDS-DMA pieces; 48 MFMAs in 8 groups fed by 40 ds_read_b128; 168 VALU interleaved 1 MFMA : 4 VALU]
//   epilogue  ~1,300 VALU (128 SiLUs x 6, row dots, per-node sums)
// MODE 0: two 256-thread workgroups per CU, 8 DMA pieces per wavefront and chunk (32 KB per workgroup: head + tail)  = today
// MODE 1: one 512-thread workgroup per CU, 2 pieces per wavefront and chunk (16 KB per CU: tails only), heads read from a resident
//         128 KB image, tiles in a persistent loop (TILES per wavefront)
// MODE 2: MODE 0 without any DMA (the ablation's "stream off")       MODE 3: MODE 1 without DMA
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
template <int G> struct IC { static constexpr int value = G; };
template <int I, int N, typename F> DEVINL void static_for(F&& f) { if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); } }

template <unsigned O0, unsigned O1, unsigned O2, unsigned O3>
DEVINL void lds_read4(f16x8 (&f)[4], unsigned addr) {
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(addr), "i"(O0), "i"(O1), "i"(O2), "i"(O3));
}
template <int N> DEVINL void lds_wait4(f16x8 (&f)[4]) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "i"(N)); }
DEVINL void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
DEVINL void split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
    const f16x2 hp = __builtin_convertvector((f32x2){y0, y1}, f16x2);
    const float l0 = y0 - (float)hp[0], l1 = y1 - (float)hp[1];
    hi = __builtin_bit_cast(uint32_t, hp);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){l0, l1}, f16x2));
}

// fragments of units 2g, 2g+1: (head, tail) of each
template <int g>
DEVINL void req4(f16x8 (&f)[4], unsigned hb, unsigned tb) {
    asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %5 offset:%7"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(hb), "v"(tb), "i"(2 * g * 1024), "i"((2 * g + 1) * 1024));
}

template <int MODE>
__global__ __launch_bounds__((MODE & 1) ? 512 : 256, (MODE & 1) ? 1 : 2) void k(float* out, const float* in, const int* meta, int tiles, int rows) {
    constexpr bool WG8 = MODE & 1, DMA = MODE < 2;
    constexpr int NW = WG8 ? 8 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // WG8: [head image 32768 floats = 128 KB][tail double buffer 2 x 4096 floats];  else [2 x 8192 floats (head + tail chunk)]
    float* resident = smem;
    float* stream = WG8 ? smem + 32768 : smem;
    __shared__ __attribute__((aligned(16))) float wrd_s[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, n = lane & 31;
    for (int i = tid; i < (WG8 ? 32768 + 8192 : 16384); i += 64 * NW) smem[i] = in[i & 4095] * 1e-3f;
    for (int i = tid; i < 1024; i += 64 * NW) wrd_s[i] = in[i] * 1e-3f;
    __syncthreads();
    const unsigned s_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)stream + lane * 16;
    const unsigned r_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)resident + lane * 16;
    float s = 0.f;
    int gc = 0;
    for (int t = 0; t < tiles; ++t) {
        // ---- prologue: dependent loads (tile metadata -> node ids -> coordinates -> rows)
        const int tile = (blockIdx.x * NW + wave) * tiles + t;
        const int mol = (tile / 27) % (rows / 30);                     // a tile's rows belong to one molecule: 30 nodes = 60 KB of AB rows (L2 / L1 hits)
        const int ni = mol * 30 + meta[(tile * 32 + n) % (rows * 29)] % 30;
        const int nj = mol * 30 + meta[ni * 7 + n] % 30;
        const f32x4 xi = *reinterpret_cast<const f32x4*>(in + (size_t)ni * 512), xj = *reinterpret_cast<const f32x4*>(in + (size_t)nj * 512 + 256);
        const float radial = (xi[0] - xj[0]) * (xi[0] - xj[0]) + 1.f, d0 = (xi[1] - xj[1]) * (xi[1] - xj[1]) + 2.f, finv = 0.5f;
        const float* Arow = in + (size_t)ni * 512 + 16 * hh;
        const float* Brow = in + (size_t)nj * 512 + 256 + 16 * hh;
        f32x4 pa[4], pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pa[u] = *reinterpret_cast<const f32x4*>(Arow + 4 * u); pb[u] = *reinterpret_cast<const f32x4*>(Brow + 4 * u); }
        u32x4 phc[2], plc[2];
        auto make_ops = [&](int c, u32x4 (&ph)[2], u32x4 (&pl)[2]) {
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
                const int cn = (c + 1) & 7;
                pa[u] = *reinterpret_cast<const f32x4*>(Arow + 32 * cn + 4 * u);
                pb[u] = *reinterpret_cast<const f32x4*>(Brow + 32 * cn + 4 * u);
            }
        };
        make_ops(0, phc, plc);
        f32x16 acc[8];
        // ---- chunk loop
#pragma unroll 1
        for (int c = 0; c < 8; ++c, ++gc) {
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            const int buf = gc & 1;
            if constexpr (DMA) {
                if constexpr (WG8) {      // tails of the next chunk: 16 KB per CU, 2 KiB per wavefront
                    const float* src = in + ((size_t)((gc + 1) & 7) * 4096 + wave * 512 + lane * 4);
                    float* dst = stream + (buf ^ 1) * 4096 + wave * 512;
                    glds16(src, dst); glds16(src + 256, dst + 256);
                } else {                  // head + tail of the next chunk: 32 KB per workgroup, 8 KiB per wavefront
                    const float* src = in + ((size_t)((gc + 1) & 7) * 8192 + wave * 2048 + lane * 4);
                    float* dst = stream + (buf ^ 1) * 8192 + wave * 2048;
#pragma unroll
                    for (int u = 0; u < 8; ++u) glds16(src + u * 256, dst + u * 256);
                }
            }
            // fragment addresses: heads [k-step][ct] and tails; WG8: heads from the resident image of chunk c, tails from the stream
            const unsigned hb = WG8 ? r_lds + c * 16384 : s_lds + buf * 32768;
            const unsigned tb = WG8 ? s_lds + buf * 16384 : s_lds + buf * 32768 + 16384;
            u32x4 phn[2], pln[2];
            f16x8 f0[4], f1[4];
            req4<0>(f0, hb, tb);
            static_for<0, 8>([&](auto Gc) {
                constexpr int g = decltype(Gc)::value;
                f16x8(&cur)[4] = (g & 1) ? f1 : f0;
                f16x8(&nxt)[4] = (g & 1) ? f0 : f1;
                lds_wait4<0>(cur);
                if constexpr (g + 1 < 8) req4<g + 1>(nxt, hb, tb);
                if constexpr (g == 1) make_ops((c + 1) & 7, phn, pln);     // the next chunk's operands ride in this chunk
                constexpr int c0 = (2 * g) & 7, c1 = (2 * g + 1) & 7, st = g >> 2;
                const f16x8 A_h = __builtin_bit_cast(f16x8, phc[st]), A_l = __builtin_bit_cast(f16x8, plc[st]);
                if (c == 0 && st == 0) {
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, cur[0], f32x16{}, 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, cur[2], f32x16{}, 0, 0, 0);
                } else {
                    acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, cur[0], acc[c0], 0, 0, 0);
                    acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, cur[2], acc[c1], 0, 0, 0);
                }
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_l, cur[0], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_l, cur[2], acc[c1], 0, 0, 0);
                acc[c0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, cur[1], acc[c0], 0, 0, 0);
                acc[c1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_h, cur[3], acc[c1], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            });
#pragma unroll
            for (int st = 0; st < 2; ++st) { phc[st] = phn[st]; plc[st] = pln[st]; }
        }
        // ---- epilogue: 128 SiLUs (un-scale, exp, +1, rcp, mul, dot) + row dots + per-node sums (~2.1 segments)
        float dot[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dot[r] = 0.f;
        const float wav = wrd_s[lane], rs = wrd_s[64 + lane], b2v = wrd_s[128 + lane];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cc][r] = __builtin_fmaf(acc[cc][r], rs, b2v);
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(acc[cc][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = 1.0f + e[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_rcpf(e[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cc][r] *= e[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) dot[r] = __builtin_fmaf(acc[cc][r], wav, dot[r]);
        }
        float rowdot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) rowdot += __shfl_xor(dot[r], 1 + (r & 15));
        const float att = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(rowdot));
        float sm[2] = {0.f, 0.f};
#pragma unroll
        for (int sgi = 0; sgi < 2; ++sgi)
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) v = __builtin_fmaf(att * (float)(r + sgi), acc[cc][r], v);
                sm[sgi] += v;
                if (hh == 0) out[((size_t)(tile & 4095) * 8 + cc) * 32 + n + sgi * 64] = v;
            }
        s += sm[0] + sm[1];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[(1 << 20) + blockIdx.x * 64 * NW + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, const float* in, const int* meta, int tiles_per_wave) {
    constexpr bool WG8 = MODE & 1;
    const size_t lds = WG8 ? (32768 + 8192) * 4 : 16384 * 4;
    CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = WG8 ? 256 : 512, threads = WG8 ? 512 : 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(threads), lds, 0, out, in, meta, tiles_per_wave, 7680);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(threads), lds, 0, out, in, meta, tiles_per_wave, 7680);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // 2048 wavefronts x tiles_per_wave tiles on 1024 SIMDs
    printf("%-86s %8.3f ms  = %6.2f us per tile and SIMD  (the shipped kernel: 14.9; 384 MFMAs alone 7.6)\n", name, ms, ms * 1e3 / (2.0 * tiles_per_wave));
}

int main() {
    float *in, *out; int* meta;
    const size_t nin = (size_t)7680 * 512;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, (2 << 20) * 4)); CK(hipMalloc(&meta, nin * 4));
    float* h = (float*)malloc(nin * 4); int* hm = (int*)malloc(nin * 4);
    srand(1);
    for (size_t i = 0; i < nin; ++i) { h[i] = (float)rand() / RAND_MAX * 2.f - 1.f; hm[i] = rand(); }
    CK(hipMemcpy(in, h, nin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(meta, hm, nin * 4, hipMemcpyHostToDevice));
    const int T = 24;
    run<0>("two 4-wave workgroups per CU, head + tail streamed (today's structure)", out, in, meta, T);
    run<1>("one 8-wave workgroup per CU, heads resident, tails streamed, lockstep per chunk, persistent", out, in, meta, T);
    run<2>("two 4-wave workgroups per CU, no stream at all", out, in, meta, T);
    run<3>("one 8-wave workgroup per CU, no stream at all", out, in, meta, T);
    run<0>("(repeat) today's structure", out, in, meta, T);
    run<1>("(repeat) 8-wave, heads resident", out, in, meta, T);
    return 0;
}
