// Shared device helpers of libhierdiff_hip: vector types, LDS-DMA copy, activations, compile-time loops,
// Philox4x32-10 + Box-Muller.  Included through kernels.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define HD_DEVINL __device__ __forceinline__

HD_DEVINL void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// same with a compile-time byte offset, which the instruction applies to BOTH the global and the LDS address: a run of
// 1 KiB pieces then needs one address / one M0 value per 4 KiB instead of one per piece
template <int OFF>
HD_DEVINL void glds16o(const void* gsrc, void* lds_dst) {
    static_assert(OFF >= 0 && OFF < 4096, "12-bit immediate");
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, OFF, 0);
}

// ----------------------------------------------------------------------------- math helpers

// x * sigmoid(x).  exp(-x) = 2^(-x*log2e) with a compensated product so the exponent argument
// keeps ~1 ulp over the whole range (v_exp_f32 and v_rcp_f32 are 1-ulp instructions).
HD_DEVINL float silu_f(float x) {
    const float L2E_HI = 1.44269502162933349609375f;   // float(log2 e)
    const float L2E_LO = 1.925962991e-8f;              // log2 e - L2E_HI
    const float LN2 = 0.693147180559945309f;
    float nx = -x;
    float t = nx * L2E_HI;
    float tlo = __builtin_fmaf(nx, L2E_HI, -t) + nx * L2E_LO;
    float e = __builtin_amdgcn_exp2f(t);
    e = e * __builtin_fmaf(tlo, LN2, 1.0f);       // stays +inf for x << 0 (an fma(e, d, e) would give NaN)
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// 1 / (1 + exp(-x)) with the same compensated exponent (~2 ulp); saturates to 0 / 1.
HD_DEVINL float sigmoid_f(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925962991e-8f, LN2 = 0.693147180559945309f;
    float nx = -x;
    float t = nx * L2E_HI;
    float tlo = __builtin_fmaf(nx, L2E_HI, -t) + nx * L2E_LO;
    float e = __builtin_amdgcn_exp2f(t) * __builtin_fmaf(tlo, LN2, 1.0f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// plain SiLU of the bf16x3 node kernel (contraction error ~1e-6 anyway): exp2(-x*log2e), 5 instructions; the
// exponent argument is off by <= |x|*1.7e-7, i.e. a relative error of that size on an already saturated value.
HD_DEVINL float silu_fast(float x) {
    float e = __builtin_amdgcn_exp2f(x * -1.44269502162933349609375f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---- scaled-domain activations of the bf16x3 edge kernel.  The host multiplies everything that feeds a SiLU /
// sigmoid of the edge model by c = -log2(e) (first edge Linear incl. bias and the two distance columns, b2, the
// attention bias), so with x' = c x
//     silu'(x') := x' * rcp(1 + exp2(x')) = c * silu(x)          sigmoid(z) = rcp(1 + exp2(z'))
// need no multiply by log2(e); the factor c carried by the activations is undone by 1/c folded into the
// weights that consume them (W2: c * 1/c = 1, i.e. unchanged; coord_mlp.4; the neighbour-sum half of node_mlp.0).
// Deliberately NOT written with v_pk_*_f32: packed fp32 runs on the matrix pipe's datapath and cannot issue while
// an MFMA of either co-resident wavefront is in flight (scratch/mb/coissue.hip: 4 v_pk_fma per MFMA cost
// 52 ns/slot vs 30 ns for 4 v_fma_f32, which hide completely), so the file is built with -fno-slp-vectorize.
HD_DEVINL float silu_scaled(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)); }
// bf16 head / tail of a pair, each packed into one dword (element 0 in the low half)
HD_DEVINL void bf16_split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
    const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){y0, y1}, bf16x2_t));
    const float l0 = y0 - __builtin_bit_cast(float, hp << 16);
    const float l1 = y1 - __builtin_bit_cast(float, hp & 0xffff0000u);
    hi = hp;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){l0, l1}, bf16x2_t));
}

// fp16 head / tail of a pair ("fp16x3", PREC 3): 11 + 11 significant bits, |y - h - l| <= 2^-22 |y| as long as the tail stays a
// normal fp16 number (|y| >= 2^-2 after the operand scaling described in k_edge.hpp); below that the tail is a SUBNORMAL fp16
// with absolute error 2^-25, which v_mfma_f32_32x32x16_f16 takes at face value on gfx950 (scratch/mb/f16_denorm.hip).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
HD_DEVINL void f16_split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
    const f16x2_t hp = __builtin_convertvector((f32x2){y0, y1}, f16x2_t);      // round to nearest even
    const float l0 = y0 - (float)hp[0];
    const float l1 = y1 - (float)hp[1];
    hi = __builtin_bit_cast(uint32_t, hp);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){l0, l1}, f16x2_t));
}
// the two-way modes share every line but these two: PREC 1 = bf16 pieces, PREC 3 = fp16 pieces
template <bool F16>
HD_DEVINL void split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
    if constexpr (F16) f16_split2(y0, y1, hi, lo);
    else bf16_split2(y0, y1, hi, lo);
}
template <bool F16>
HD_DEVINL f32x16 mma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// fp16 mode: floor of the per-edge bound on the first-layer activation (keeps the scale's exponent arithmetic in range)
#define HD_F16_FLOOR 9.5367431640625e-07f      // 2^-20
#define HD_TWOWAY(p) ((p) == 1 || (p) == 3)

// three-way split (head, middle, tail: 24 significant bits, |y - h - m - l| <= 2^-27 |y|) for the bf16x6 contraction
HD_DEVINL void bf16_split3(float y0, float y1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){y0, y1}, bf16x2_t));
    const float r0 = y0 - __builtin_bit_cast(float, hp << 16);
    const float r1 = y1 - __builtin_bit_cast(float, hp & 0xffff0000u);
    const uint32_t mp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, bf16x2_t));
    const float s0 = r0 - __builtin_bit_cast(float, mp << 16);
    const float s1 = r1 - __builtin_bit_cast(float, mp & 0xffff0000u);
    hi = hp;
    mid = mp;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){s0, s1}, bf16x2_t));
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0..N-1
template <int I, int N, typename F>
HD_DEVINL void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ----------------------------------------------------------------------------- Philox4x32-10

HD_DEVINL void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// normal(seed, sample, draw, index): counter = (index/2, draw, sample.lo, sample.hi), key = seed.
// Each counter block yields two Box-Muller normals; index & 1 selects one.
HD_DEVINL float philox_normal(uint64_t seed, uint64_t sample, uint32_t draw, uint32_t index) {
    uint32_t c0 = index >> 1, c1 = draw, c2 = (uint32_t)sample, c3 = (uint32_t)(sample >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    float u1 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float u2 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float rad = sqrtf(-2.0f * logf(u1));
    float ang = 6.283185307179586f * u2;
    return (index & 1) ? rad * sinf(ang) : rad * cosf(ang);
}
