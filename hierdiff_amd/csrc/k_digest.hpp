// Content digest of a set of parameter tensors (round 6; no reference counterpart - the reference evaluates its nn.Modules in
// place, this library evaluates PACKED IMAGES of them, and an image must never outlive the values it was packed from).
//
// digest = sum over the PAIRS (w[i], w[i+1]), i even, of the virtual concatenation of the tensors' 32-bit words (padded with zeros to a
// multiple of four) of splitmix64((w[i] | w[i+1] << 32) + phi64 * (i + 1)), modulo 2^64.
// Integer addition is associative, so the value does not depend on the launch geometry or on the order the atomics land in; it
// changes when any single bit of any word changes (splitmix64 is a bijection of the 64-bit argument, and the argument differs),
// and two different contents collide with probability 2^-64.  HBM/L2-bound: 16-byte loads where a group of four words is aligned
// and inside one tensor, 23.7 MB at L = 6.  One launch, no memset, no copy: every workgroup adds its partial sum to `state[0]` and
// takes a ticket from `state[1]`; the last one publishes the total to `host_out` (pinned, mapped host memory) and leaves both
// words zero for the next call.
#pragma once
#include "common.hpp"

struct DigestArgs {
    const uint32_t* const* ptrs;        // [n] device pointers (4-byte aligned)
    const long long* prefix;            // [n + 1] word offsets of the tensors in the concatenation
    int n;
    long long total;
    unsigned long long* state;          // {accumulator, tickets}: zero on entry, zero again on exit
    unsigned long long* host_out;       // pinned host word (device-visible address)
};

__device__ __forceinline__ unsigned long long digest_mix(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ unsigned long long digest_term(unsigned long long w, long long i) {
    return digest_mix(w + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
}

constexpr int DIGEST_CHUNK = 16384;     // words per workgroup (64 KB)

__global__ __launch_bounds__(256) void k_params_digest(DigestArgs a) {
    __shared__ unsigned long long wsum[4];
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
    for (int k = 4 * threadIdx.x; k < DIGEST_CHUNK; k += 4 * 256) {        // four consecutive words per lane and step
        const long long i = base + k;
        if (i >= a.total) break;
        while (i >= t_end) {             // the chunk crosses into the next tensor(s); empty tensors are skipped by the same loop
            ++ti;
            t_begin = a.prefix[ti]; t_end = a.prefix[ti + 1];
            p = a.ptrs[ti];
        }
        const uint32_t* q = p + (i - t_begin);
        if (i + 4 <= t_end && (reinterpret_cast<uintptr_t>(q) & 15) == 0) {
            const uint4 v = *reinterpret_cast<const uint4*>(q);
            acc += digest_term((unsigned long long)v.x | ((unsigned long long)v.y << 32), i) +
                   digest_term((unsigned long long)v.z | ((unsigned long long)v.w << 32), i + 2);
        } else {                         // tensor boundary or an unaligned view: word by word, walking on as needed
            int tj = ti;
            long long b = t_begin, e = t_end;
            const uint32_t* pj = p;
            uint32_t w[4] = {0u, 0u, 0u, 0u};        // (words past the end of the concatenation count as zero: `total` is part of the key)
            for (int d = 0; d < 4 && i + d < a.total; ++d) {
                while (i + d >= e) { ++tj; b = a.prefix[tj]; e = a.prefix[tj + 1]; pj = a.ptrs[tj]; }
                w[d] = pj[i + d - b];
            }
            acc += digest_term((unsigned long long)w[0] | ((unsigned long long)w[1] << 32), i) +
                   digest_term((unsigned long long)w[2] | ((unsigned long long)w[3] << 32), i + 2);
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (s) atomicAdd(&a.state[0], s);
        __threadfence();
        const unsigned long long ticket = atomicAdd(&a.state[1], 1ull);
        if (ticket == (unsigned long long)gridDim.x - 1) {       // every other workgroup's sum is in state[0] (fence before its ticket)
            __threadfence();
            const unsigned long long tot = atomicExch(&a.state[0], 0ull);
            atomicExch(&a.state[1], 0ull);
            __hip_atomic_store(a.host_out, tot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
