// General exact-fp32 GEMM of the TRAINING path (k_tgemm): the node-level Linears around an edge layer, forward and both
// backward directions, and the dense reduction over all edge rows dW2 = G2^T P (hierdiff_amd/training.py; reference:
// the nn.Linear modules of egnn_new.py:9-33,74-89 under torch.autograd, diffusion_qm9.py:774-777).  Included through kernels.hpp.
//
//     C[m][n] = epi( sum_k A(m, k) B(k, n) ),   A(m, k) = A[m sam + k sak],   B(k, n) = B[k sbk + n sbn]
//
// with one unit stride per operand, so that one kernel serves
//     forward   Y  = X W^T + b     A = X  (k contiguous),  B(k, n) = W[n][k] (k contiguous)
//     backward  dX = dY W          A = dY (k contiguous),  B(k, n) = W[k][n] (n contiguous)
//     backward  dW = dY^T X        A(m, k) = dY[k][m] (m contiguous),  B(k, n) = X[k][n] (n contiguous), k = node / edge rows,
//                                  cut into gridDim.z slabs of kslab rows (split-K: per-slab partial results in `ws`, summed in
//                                  slab order by k_tgemm_reduce - deterministic, no atomics), the bias gradient
//                                  db[m] = sum_k dY[k][m] riding along as column sums of the A tiles.
// Workgroup tile 64 x 128, K chunks of 32 double-buffered through LDS as [row][k] (row stride 36 floats: a lane's four k
// values are one conflict-free ds_read_b128 whatever the source layout; m- / n-contiguous sources are transposed on the LDS
// write, conflict-free through a quad swizzle), four wavefronts of 32 x 64 (two v_mfma_f32_32x32x2_f32 accumulators), global
// loads two to three chunks ahead in registers, the LDS fragments of the next chunk requested before the MFMAs of the
// current one, one barrier per chunk, two workgroups per CU (55 KB of LDS).  Sources that are not 16-byte aligned (the embedding Linears, K = 9
// or 10) take scalar loads.  Exact fp32 like every other kernel of the training path; the order of the k sum differs from
// the BLAS library's, i.e. results agree with it to fp32 round-off.
#pragma once
#include "common.hpp"

enum { TG_EPI_BIAS = 0, TG_EPI_BIAS_SILU2 = 1, TG_EPI_RESID_MASK = 2, TG_EPI_MUL_DSILU = 3 };

struct TGemmArgs {
    const float* A;
    const float* B;
    float* C;               // [M][ldc]
    float* C2;              // BIAS_SILU2: silu(C)
    const float* bias;      // [N] or null
    const float* aux;       // RESID_MASK: residual R [M][ldc]; MUL_DSILU: pre-activation [M][ldc]
    const float* rmask;     // RESID_MASK: [M] row mask or null
    float* ws;              // split-K: [z][M][N] partial results (null: one slab, epilogue applied here)
    float* colsum_ws;       // non-null: [z][M] partial sums over k of A(m, k) (written by the n-tile-0 workgroups)
    float* colsum;          // k_tgemm_small: [M] the column sums themselves
    long long sam, sak, sbk, sbn;
    int M, N, K, ldc, kslab, epi, avec, bvec;
    int nx, ny, nz;         // nz > 0: 1-D XCD-aware launch of nx x ny tiles x nz slabs (split-K)
};

constexpr int TG_BM = 64, TG_BN = 128, TG_BK = 32;

// d/dx [x sigmoid(x)] = s (1 + x (1 - s))
HD_DEVINL float dsilu_f(float x) {
    const float s = sigmoid_f(x);
    return s * __builtin_fmaf(x, 1.0f - s, 1.0f);
}

// FAST: every tile is full and every row 16-byte aligned (M % 64 == 0, N % 128 == 0, slabs of whole chunks) - all node-level
// and edge-row shapes of the production widths.  Its loads carry no bounds checks and no branches: hipcc assumes that
// nothing is in flight behind a control-flow join and answers a load inside an `if` with s_waitcnt vmcnt(0) at the join, which
// exposes the full L2 / HBM latency once per chunk (measured: 35 us for a 7,680 x 256 x 256 product whichever prefetch
// distance was written down).  The general form (embedding Linears with K = 9 / 10 columns, ragged row counts) keeps the checks.
template <bool A_KC, bool B_KC, bool FAST>
__global__ __launch_bounds__(256, 2) void k_tgemm(TGemmArgs g) {
    // LDS: [row][k] with k contiguous, row stride 36 floats (conflict-free ds_read_b128 of a lane's four k values)
    constexpr int LDK = TG_BK + 4;
    __shared__ __attribute__((aligned(16))) float As[2][TG_BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][TG_BN * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, hh = lane >> 5, n = lane & 31;
    // Split-K launches are 1-D and XCD-aware (workgroup id % 8 = XCD): the tiles of one K slab - which read the same rows of
    // both operands - run on ONE XCD and share its L2 (a plain 3-D grid deals the eight 64 x 128 tiles of a 256 x 256 result
    // to eight different XCDs: every slab is then fetched from HBM / Infinity Cache up to four times).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.nz > 0) {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3, tiles = g.nx * g.ny, t = j % tiles;
        bz = (j / tiles) * 8 + xcd;
        if (bz >= g.nz) return;
        bx = t % g.nx; by = t / g.nx;
    }
    const int m0 = bx * TG_BM, n0 = by * TG_BN;
    const int kbeg = bz * g.kslab, kend = min(g.K, kbeg + g.kslab);
    const int nchunk = (kend - kbeg + TG_BK - 1) / TG_BK;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    // one float4 of a tile: `row` along the operand's strided axis, `x` along its contiguous one (limit xend)
    auto ld4 = [&](const float* base, long long stride, int row, int x, int xend, bool vec) -> f32x4 {
        const float* p = base + (long long)row * stride + x;
        if constexpr (FAST) return *reinterpret_cast<const f32x4*>(p);
        f32x4 v = z4;
        if (vec && x + 3 < xend) v = *reinterpret_cast<const f32x4*>(p);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (x + j < xend) v[j] = p[j];
        }
        return v;
    };
    // ---- one 16-byte piece of a chunk (pieces 0, 1: the A tile; 2 .. 5: the B tile), global -> registers
    auto load_piece = [&](int c, int u, f32x4& r) {
        const int k0 = kbeg + c * TG_BK;
        if (u < 2) {
            const int idx = tid + 256 * u;
            if constexpr (!FAST) r = z4;
            if constexpr (A_KC) {
                const int m = m0 + (idx >> 3), k = k0 + 4 * (idx & 7);
                if (FAST || (m < g.M && k < kend)) r = ld4(g.A, g.sam, m, k, kend, g.avec);
            } else {
                const int k = k0 + (idx >> 4), m = m0 + 4 * (idx & 15);
                if (FAST || (k < kend && m < g.M)) r = ld4(g.A, g.sak, k, m, g.M, g.avec);
            }
        } else {
            const int idx = tid + 256 * (u - 2);
            if constexpr (!FAST) r = z4;
            if constexpr (B_KC) {
                const int nn = n0 + (idx >> 3), k = k0 + 4 * (idx & 7);
                if (FAST || (nn < g.N && k < kend)) r = ld4(g.B, g.sbn, nn, k, kend, g.bvec);
            } else {
                const int k = k0 + (idx >> 5), nn = n0 + 4 * (idx & 31);
                if (FAST || (k < kend && nn < g.N)) r = ld4(g.B, g.sbk, k, nn, g.N, g.bvec);
            }
        }
    };
    // ---- registers -> LDS: k-contiguous sources as they come (one ds_write_b128), m- / n-contiguous ones transposed.
    // The eight k quads of a row are permuted by sw(row) = (row >> 4) & 3 (quad ^ sw): the transposing writes of the 16
    // lanes that share a k then hit 16 different banks (16 (row/4 & 3) + 4 (quad ^ sw)) instead of four, and the
    // ds_read_b128 of 16 consecutive rows stay conflict-free (sw is constant over them).
    auto sw = [](int row) { return (row >> 4) & 3; };
    auto store_piece = [&](int buf, int u, const f32x4& v) {
        if (u < 2) {
            const int idx = tid + 256 * u;
            if constexpr (A_KC) {
                const int row = idx >> 3;
                *reinterpret_cast<f32x4*>(&As[buf][row * LDK + 4 * ((idx & 7) ^ sw(row))]) = v;
            } else {
                const int k = idx >> 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 4 * (idx & 15) + j;
                    As[buf][row * LDK + 4 * ((k >> 2) ^ sw(row)) + (k & 3)] = v[j];
                }
            }
        } else {
            const int idx = tid + 256 * (u - 2);
            if constexpr (B_KC) {
                const int row = idx >> 3;
                *reinterpret_cast<f32x4*>(&Bs[buf][row * LDK + 4 * ((idx & 7) ^ sw(row))]) = v;
            } else {
                const int k = idx >> 5;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 4 * (idx & 31) + j;
                    Bs[buf][row * LDK + 4 * ((k >> 2) ^ sw(row)) + (k & 3)] = v[j];
                }
            }
        }
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // lane (row n, half hh) feeds k = 16 hh + 4 q + j to instruction (q, j) of a chunk - the same map for both operands.
    // A chunk's twelve fragments: piece r = 3 q + {0: A, 1: B columns 0-31, 2: B columns 32-63 of the wavefront's 64}.
    struct Frag { f32x4 v[12]; };
    const int arow = 32 * wr + n, brow = 64 * wc + n;
    auto read_frag = [&](int buf, int r, f32x4& dst) {
        const int q = r / 3, w = r % 3;
        const int row = w == 0 ? arow : (w == 1 ? brow : brow + 32);
        const float* base = w == 0 ? &As[buf][row * LDK] : &Bs[buf][row * LDK];
        dst = *reinterpret_cast<const f32x4*>(base + 4 * ((4 * hh + q) ^ sw(row)));
    };
    f32x4 csum = z4;                                        // colsum_ws: this thread's share of sum_k A(m, k) (A m-contiguous)
    const bool do_colsum = !A_KC && g.colsum_ws != nullptr && by == 0;

    // Pipeline: one barrier per chunk; LDS buffers, fragment sets and global staging sets all of period two (the loop is
    // unrolled by two).  Step c issues, BETWEEN the 32 MFMAs of chunk c (fragment set c&1, filled during step c-1):
    //   the six global loads of chunk c+3 (staging set (c+1)&1, whose chunk c+1 went to LDS during step c-1),
    //   the twelve fragment reads of chunk c+1 from buffer (c+1)&1 (stored during step c-1, published by the barrier),
    //   the six LDS stores of chunk c+2 (staging set c&1, requested during step c-1) into buffer c&1 - last read for the
    //   fragments of chunk c, which every wavefront holds in registers since before the barrier that ended step c-1.
    // Every memory instruction is issued in the 64-cycle shadow of an MFMA (a sched_barrier after each pair pins the order:
    // left alone hipcc sinks the LDS reads next to their uses - i.e. behind the barrier, where all eight wavefronts of the CU
    // then queue at the LDS with the matrix pipe idle: MFMA-busy 0.51 - and hoists the stores in front of the MFMAs: 0.62).
    // No load sits behind a branch: chunk indices past the slab are clamped (they re-read / re-store the last chunk, whose
    // copy is never used).
    f32x4 gs[2][6];
    Frag fr[2];
    const int last = max(nchunk - 1, 0);
    auto step = [&](auto Par, int c) {
        constexpr int P = decltype(Par)::value;             // parity of c
        const int cl = min(c + 3, last);
        static_for<0, 32>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            constexpr int q = i / 8, j = (i / 2) % 4;
            if constexpr ((i & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[P].v[3 * q][j], fr[P].v[3 * q + 1][j], acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[P].v[3 * q][j], fr[P].v[3 * q + 2][j], acc1, 0, 0, 0);
            if constexpr (i < 6) load_piece(cl, i, gs[P ^ 1][i]);
            else if constexpr (i < 18) read_frag(P ^ 1, i - 6, fr[P ^ 1].v[i - 6]);
            else if constexpr (i >= 20 && i < 26) store_piece(P, i - 20, gs[P][i - 20]);
            if constexpr (i < 26) __builtin_amdgcn_sched_barrier(0);
        });
        if (do_colsum && c + 2 < nchunk) csum += gs[P][0] + gs[P][1];
        __syncthreads();
    };
    if (nchunk > 0) {
        // chunks 0 and 1 to LDS, chunk 2 requested, fragments of chunk 0 in registers
        static_for<0, 6>([&](auto Uc) { load_piece(0, decltype(Uc)::value, gs[0][decltype(Uc)::value]); });
        static_for<0, 6>([&](auto Uc) { load_piece(min(1, last), decltype(Uc)::value, gs[1][decltype(Uc)::value]); });
        static_for<0, 6>([&](auto Uc) { store_piece(0, decltype(Uc)::value, gs[0][decltype(Uc)::value]); });
        if (do_colsum) csum += gs[0][0] + gs[0][1];
        static_for<0, 6>([&](auto Uc) { store_piece(1, decltype(Uc)::value, gs[1][decltype(Uc)::value]); });
        if (do_colsum && 1 < nchunk) csum += gs[1][0] + gs[1][1];
        static_for<0, 6>([&](auto Uc) { load_piece(min(2, last), decltype(Uc)::value, gs[0][decltype(Uc)::value]); });
        __syncthreads();
        static_for<0, 12>([&](auto Rc) { read_frag(0, decltype(Rc)::value, fr[0].v[decltype(Rc)::value]); });
        __syncthreads();                                    // everybody holds its fragments of chunk 0: step 0 may overwrite buffer 0
    }
    for (int c = 0; c < nchunk; c += 2) {
        step(std::integral_constant<int, 0>{}, c);
        if (c + 1 >= nchunk) break;
        step(std::integral_constant<int, 1>{}, c + 1);
    }

    if (do_colsum) {                                        // sum over the 16 thread rows of the loader grid, in row order
        float* red = &Bs[0][0];                             // [16][64]; the chunk loop ended with a barrier
        *reinterpret_cast<f32x4*>(red + (tid >> 4) * 64 + 4 * (tid & 15)) = csum;
        __syncthreads();
        if (tid < 64 && m0 + tid < g.M) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * 64 + tid];
            g.colsum_ws[(size_t)bz * g.M + m0 + tid] = s;
        }
    }

    // ---- epilogue.  acc[r]: row m0 + 32 wr + rho(r), rho(r) = (r & 3) + 8 (r >> 2) + 4 hh; column n0 + 64 wc + 32 cn + n
    const bool split = g.ws != nullptr;
#pragma unroll
    for (int cn = 0; cn < 2; ++cn) {
        const int col = n0 + 64 * wc + 32 * cn + n;
        if (col >= g.N) continue;
        const float bias = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wr + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (row >= g.M) continue;
            float v = (cn == 0 ? acc0[r] : acc1[r]);
            if (split) { g.ws[((size_t)bz * g.M + row) * g.N + col] = v; continue; }
            v += bias;
            const size_t o = (size_t)row * g.ldc + col;
            if (g.epi == TG_EPI_BIAS_SILU2) { g.C[o] = v; g.C2[o] = silu_f(v); }
            else if (g.epi == TG_EPI_RESID_MASK) { v += g.aux[o]; if (g.rmask) v *= g.rmask[row]; g.C[o] = v; }
            else if (g.epi == TG_EPI_MUL_DSILU) g.C[o] = v * dsilu_f(g.aux[o]);
            else g.C[o] = v;
        }
    }
}

// Small results (round 5): when the 64 x 128 tiling gives the chip fewer than ~128 workgroups - the node-level Linears of a small
// training batch: 480 rows at the reference's batch size of 16 are 16-32 workgroups, ~25 us per GEMM whatever its size - the same
// product runs as 32 x 32 output tiles, one workgroup each, the four wavefronts taking the four QUARTERS OF K: operands straight from
// global memory / L2 into the MFMA operand registers (a wave's share is 32 rows x K/4: no LDS staging, next step's loads in flight under
// the current step's four MFMAs), the four partial tiles added through LDS in wave order (deterministic), every epilogue of k_tgemm,
// and the bias gradient (column sums of an m-contiguous A) on the way.  Lane (n, hh) of a step of eight k values holds k0 + 4 hh + j
// for MFMA j of both operands - a float4 along a k-contiguous source, four coalesced loads along an m- / n-contiguous one.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_tgemm_small(TGemmArgs g) {
    __shared__ float red[4][16 * 64];
    __shared__ float cred[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, n = lane & 31;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int row = m0 + n, col = n0 + n;
    const int kq = ((g.K + 3) / 4 + 7) / 8 * 8;                 // K quarter, whole steps of eight
    const int kbeg = wave * kq, kend = min(g.K, kbeg + kq);
    const bool rok = row < g.M, cok = col < g.N;
    const float* Ap = A_KC ? g.A + (long long)row * g.sam : g.A + row;
    const float* Bp = B_KC ? g.B + (long long)col * g.sbn : g.B + col;
    auto load = [&](int k0, f32x4& a, f32x4& b) {
        const int k = k0 + 4 * hh;
        a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a;
        if (rok) {
            if (A_KC) {
                if (g.avec && k + 3 < kend) a = *reinterpret_cast<const f32x4*>(Ap + k);
                else { for (int j = 0; j < 4; ++j) if (k + j < kend) a[j] = Ap[k + j]; }
            } else { for (int j = 0; j < 4; ++j) if (k + j < kend) a[j] = Ap[(long long)(k + j) * g.sak]; }
        }
        if (cok) {
            if (B_KC) {
                if (g.bvec && k + 3 < kend) b = *reinterpret_cast<const f32x4*>(Bp + k);
                else { for (int j = 0; j < 4; ++j) if (k + j < kend) b[j] = Bp[k + j]; }
            } else { for (int j = 0; j < 4; ++j) if (k + j < kend) b[j] = Bp[(long long)(k + j) * g.sbk]; }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float csum = 0.f;
    f32x4 a0, b0, a1, b1;
    if (kbeg < kend) load(kbeg, a0, b0);
    for (int k0 = kbeg; k0 < kend; k0 += 8) {
        if (k0 + 8 < kend) load(k0 + 8, a1, b1);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc, 0, 0, 0);
        csum += (a0[0] + a0[1]) + (a0[2] + a0[3]);
        a0 = a1; b0 = b1;
    }
    // partial tiles -> LDS [wave][r][lane]; acc[r]: row m0 + rho(r) (+ 4 hh), column n0 + n
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
    const bool do_colsum = !A_KC && g.colsum != nullptr && blockIdx.y == 0;
    if (do_colsum) {
        csum += __shfl_xor(csum, 32);
        if (hh == 0) cred[wave][n] = csum;
    }
    __syncthreads();
    if (do_colsum && tid < 32 && m0 + tid < g.M) g.colsum[m0 + tid] = (cred[0][tid] + cred[1][tid]) + (cred[2][tid] + cred[3][tid]);
    const int l2 = tid & 63, h2 = l2 >> 5, n2 = l2 & 31, c2 = n0 + n2;
    if (c2 >= g.N) return;
    const float bias = g.bias ? g.bias[c2] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (tid >> 6) * 4 + q;
        const int rw = m0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
        if (rw >= g.M) continue;
        float v = ((red[0][r * 64 + l2] + red[1][r * 64 + l2]) + red[2][r * 64 + l2]) + red[3][r * 64 + l2];
        v += bias;
        const size_t o = (size_t)rw * g.ldc + c2;
        if (g.epi == TG_EPI_BIAS_SILU2) { g.C[o] = v; g.C2[o] = silu_f(v); }
        else if (g.epi == TG_EPI_RESID_MASK) { v += g.aux[o]; if (g.rmask) v *= g.rmask[rw]; g.C[o] = v; }
        else if (g.epi == TG_EPI_MUL_DSILU) g.C[o] = v * dsilu_f(g.aux[o]);
        else g.C[o] = v;
    }
}

// C[m][n] = sum_z ws[z][m][n] (+ bias[n]), slabs in ascending order; blocks beyond the matrix finish the column sums:
// colsum[m] = sum_z colsum_ws[z][m]
struct TGemmReduceArgs {
    const float* ws;
    const float* colsum_ws;
    const float* bias;
    float* C;
    float* colsum;
    int M, N, ldc, nz;
};

__global__ __launch_bounds__(256) void k_tgemm_reduce(TGemmReduceArgs g) {
    const long long total = (long long)g.M * g.N;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) {
        float s = 0.f;
        for (int z = 0; z < g.nz; ++z) s += g.ws[(size_t)z * total + i];
        const int m = (int)(i / g.N), n = (int)(i - (long long)m * g.N);
        if (g.bias) s += g.bias[n];
        g.C[(size_t)m * g.ldc + n] = s;
    } else if (g.colsum && i - total < g.M) {
        const int m = (int)(i - total);
        float s = 0.f;
        for (int z = 0; z < g.nz; ++z) s += g.colsum_ws[(size_t)z * g.M + m];
        g.colsum[m] = s;
    }
}

// Column sums of up to four [rows][width] arrays in two launches (the per-tile partial sums hd_edge_layer_backward leaves behind:
// db2, d(w_a), d(w_r) / d(w_d), d(b_a)): stage 1 sums CS_CHUNKS row ranges into ws [CS_CHUNKS][total width], stage 2 adds the
// ranges in ascending order - deterministic.
constexpr int CS_CHUNKS = 32;
struct ColSumArgs {
    const float* src[4];
    float* dst[4];
    int width[4], off[5];      // off[i] = first global column of array i; off[n] = total width
    float* ws;
    int rows, n;
};

__global__ __launch_bounds__(256) void k_colsum_stage1(ColSumArgs a) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int total = a.off[a.n];
    const int per = (a.rows + CS_CHUNKS - 1) / CS_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = min(a.rows, r0 + per);
    float s = 0.f;
    if (col < total) {
        int i = 0;
        while (i + 1 < a.n && col >= a.off[i + 1]) ++i;
        const float* p = a.src[i] + (col - a.off[i]);
        const int w = a.width[i];
        // eight rows in flight per thread (one load per iteration is a chain of L2 / HBM round trips: 26 us for the 28 MB of a
        // B = 256 edge layer); eight interleaved running sums, added in a fixed order
        float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] += p[(size_t)(r + 4 * k) * w];
        }
        for (int k = 0; r < r1; r += 4, ++k) q[k] += p[(size_t)r * w];
        s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < total) a.ws[(size_t)blockIdx.y * total + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_colsum_stage2(ColSumArgs a) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int total = a.off[a.n];
    if (col >= total) return;
    float s = 0.f;
    for (int c = 0; c < CS_CHUNKS; ++c) s += a.ws[(size_t)c * total + col];
    int i = 0;
    while (i + 1 < a.n && col >= a.off[i + 1]) ++i;
    a.dst[i][col - a.off[i]] = s;
}
