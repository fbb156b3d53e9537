// Content digest of a set of parameter tensors (round 6; no reference counterpart - the reference evaluates its nn.Modules in
// place, this library evaluates PACKED IMAGES of them, and an image must never outlive the values it was packed from).
//
// digest = sum over the virtual concatenation of the tensors' 32-bit words of splitmix64(word + phi64 * (position + 1)), modulo 2^64.
// Integer addition is associative, so the value does not depend on the launch geometry or on the order the atomics land in; it
// changes when any single bit of any word changes (splitmix64 is a bijection of the 64-bit argument, and the argument differs),
// and two different contents collide with probability 2^-64.  HBM-bound: one coalesced 4-byte load per word, 23.7 MB at L = 6.
#pragma once
#include "common.hpp"

struct DigestArgs {
    const uint32_t* const* ptrs;        // [n] device pointers (4-byte aligned)
    const long long* prefix;            // [n + 1] word offsets of the tensors in the concatenation
    int n;
    long long total;
    unsigned long long* out;            // zeroed by the caller
};

__device__ __forceinline__ unsigned long long digest_mix(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

constexpr int DIGEST_CHUNK = 8192;      // words per workgroup

__global__ __launch_bounds__(256) void k_params_digest(DigestArgs a) {
    const long long base = (long long)blockIdx.x * DIGEST_CHUNK;
    // the tensor that holds the chunk's first word (binary search over <= a few hundred prefix entries, once per workgroup)
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.prefix[mid] <= base) lo = mid; else hi = mid - 1;
    }
    int ti = lo;
    long long t_begin = a.prefix[ti], t_end = a.prefix[ti + 1];
    const uint32_t* p = a.ptrs[ti];
    unsigned long long acc = 0;
    for (int k = threadIdx.x; k < DIGEST_CHUNK; k += 256) {
        const long long i = base + k;
        if (i >= a.total) break;
        while (i >= t_end) {             // chunk crosses into the next tensor(s); empty tensors are skipped by the same loop
            ++ti;
            t_begin = a.prefix[ti]; t_end = a.prefix[ti + 1];
            p = a.ptrs[ti];
        }
        const unsigned long long w = p[i - t_begin];
        acc += digest_mix(w + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(a.out, acc);
}
