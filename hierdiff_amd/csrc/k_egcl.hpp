// Stage-2 layer E_GCL (/root/reference/models/egnn/gcl.py:9-205): the row-wise kernels between its dense contractions.
// Included through kernels.hpp.
//
// Unlike the stage-1 GCL, this layer carries H-wide EDGE features from layer to layer, so its first edge Linear has a
// per-edge H x H term and the edge features are read and written once per layer ([E][H] fp32, E = bs * n^2 for the
// dense "full" graph of Edge_denoise).  The layer is run as a short sequence of fp32 MFMA GEMMs over edge rows / node
// rows (k_gemm, k_node.hpp) with these kernels in between; sums over incoming edges use a CSR of the receiving index
// (ascending edge order: deterministic).  Batches here are beam-search sized, so the
// sequence favours simplicity over fusion.
#pragma once
#include "common.hpp"

// radial = |x_row - x_col|^2, cdiff = (x_row - x_col) / (sqrt(radial + 1e-8) + 1)          (gcl.py:198-205)
// pre1 = A[row] + B[col] + radial w_r + (T1[e] | sum_d ea[e][d] w_e[d]) + sum_c ctx[row][c] w_c[c];   P = SiLU(pre1)
struct EgclPreArgs {
    const float* AB;        // [M][2H]: cols < H: W1a h + b1;  cols >= H: W1b h
    const float* T1;        // [E][H] = edge_attr W1e^T (wide edge attributes) or NULL
    const float* ea;        // [E][De] (narrow edge attributes, De < 32) or NULL
    const float* w_e;       // [De][H] columns of mes_mlp.0 for the narrow edge attributes
    const float* w_r;       // [H]
    const float* w_c;       // [ctx][H]
    const float* hin;       // [M][H] (context = its last ctx columns, the reference's slicing)
    const float* x;         // [M][4]
    const int* row;
    const int* col;
    float* P;               // [E][H]
    float* geo;             // [E][4] = {cdiff_x, cdiff_y, cdiff_z, radial}
    int E, H, De, ctx, geo_mode;      // geo_mode: the message model's distance input is 1 / radial^2 (gcl.py:170-175)
    int xs;                           // floats per coordinate row of `x` (4: the padded copy; 3: the caller's tensor, round 5)
};

__global__ void k_egcl_pre(EgclPreArgs a) {
    const int q = a.H >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = (int)(idx / q), c4 = (int)(idx - (long long)e * q);
    if (e >= a.E) return;
    const int r = a.row[e], c = a.col[e];
    const float* xr = a.x + (size_t)r * a.xs;
    const float* xc = a.x + (size_t)c * a.xs;
    const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dz = xr[2] - xc[2];
    const float radial = dx * dx + dy * dy + dz * dz;
    if (c4 == 0) {
        const float inv = 1.0f / (sqrtf(radial + 1e-8f) + 1.0f);
        *reinterpret_cast<f32x4*>(a.geo + (size_t)e * 4) = f32x4{dx * inv, dy * inv, dz * inv, radial};
    }
    const int k = 4 * c4;
    f32x4 pre = *reinterpret_cast<const f32x4*>(a.AB + (size_t)r * 2 * a.H + k) +
                *reinterpret_cast<const f32x4*>(a.AB + (size_t)c * 2 * a.H + a.H + k);
    const f32x4 wr = *reinterpret_cast<const f32x4*>(a.w_r + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(a.geo_mode ? 1.0f / (radial * radial) : radial, wr[j], pre[j]);
    if (a.T1) pre += *reinterpret_cast<const f32x4*>(a.T1 + (size_t)e * a.H + k);
    for (int d = 0; d < (a.T1 ? 0 : a.De); ++d) {
        const float v = a.ea[(size_t)e * a.De + d];
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.w_e + (size_t)d * a.H + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(v, w[j], pre[j]);
    }
    for (int d = 0; d < a.ctx; ++d) {
        const float v = a.hin[(size_t)r * a.H + a.H - a.ctx + d];
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.w_c + (size_t)d * a.H + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre[j] = __builtin_fmaf(v, w[j], pre[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) pre[j] = silu_f(pre[j]);
    *reinterpret_cast<f32x4*>(a.P + (size_t)e * a.H + k) = pre;
}

// One wavefront per edge row.  MODE 0 (gate):  ef = M (* sigmoid(wa.M + ba)) * edge_mask      (gcl.py:99-107), in place
//                              MODE 1 (coord): phi = w.C1;  trans = cdiff (tanh(phi) range | phi) edge_mask   (:130-137)
struct EgclRowArgs {
    float* X;               // [E][H] rows (MODE 0: updated in place)
    const float* w;         // [H]
    const float* bias;      // [1] or NULL
    const float* emask;     // [E] or NULL
    const float* geo;       // MODE 1: [E][4]
    float* trans;           // MODE 1: [E][4]
    float range;
    int E, H, attention, use_tanh;
};

template <int MODE>
__global__ void k_egcl_row(EgclRowArgs a) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= a.E) return;
    float* rowp = a.X + (size_t)e * a.H;
    const float m = a.emask ? a.emask[e] : 1.0f;
    float dot = 0.f;
    if (MODE == 1 || a.attention) {
        for (int k = lane * 4; k < a.H; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + k);
            const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) dot = __builtin_fmaf(v[j], w[j], dot);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    }
    if constexpr (MODE == 0) {
        const float s = (a.attention ? sigmoid_f(dot + (a.bias ? a.bias[0] : 0.0f)) : 1.0f) * m;
        for (int k = lane * 4; k < a.H; k += 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(rowp + k);
            *reinterpret_cast<f32x4*>(rowp + k) = v * s;
        }
    } else {
        if (lane == 0) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.geo + (size_t)e * 4);
            const float sc = (a.use_tanh ? tanhf(dot) * a.range : dot) * m;
            *reinterpret_cast<f32x4*>(a.trans + (size_t)e * 4) = f32x4{g[0] * sc, g[1] * sc, g[2] * sc, 0.f};
        }
    }
}

// edge_mlp first layer epilogue: E1 = SiLU(E1 + radial w_er)  (the radial column of edge_mlp.0, gcl.py:111-112); and the
// final masking of the new edge attributes, EA *= edge_mask (:114-116, :192-193).  MODE 0 / 1.
struct EgclEwArgs {
    float* X;               // [E][H]
    const float* w;         // MODE 0: [H]
    const float* geo;       // MODE 0: [E][4] (radial in .w)
    const float* emask;     // MODE 1: [E]
    int E, H;
};

template <int MODE>
__global__ void k_egcl_ew(EgclEwArgs a) {
    const int q = a.H >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int e = (int)(idx / q), c4 = (int)(idx - (long long)e * q);
    if (e >= a.E) return;
    f32x4 v = *reinterpret_cast<f32x4*>(a.X + (size_t)e * a.H + 4 * c4);
    if constexpr (MODE == 0) {
        const float radial = a.geo[(size_t)e * 4 + 3];
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.w + 4 * c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(__builtin_fmaf(radial, w[j], v[j]));
    } else {
        const float m = a.emask[e];
        v = v * m;
    }
    *reinterpret_cast<f32x4*>(a.X + (size_t)e * a.H + 4 * c4) = v;
}

// node-side packing: hin[M][H] = h[:, :H] of the [M][H+ctx] input, x4 = (x, 0), hres = hin (residual accumulator)
struct EgclNodeInArgs {
    const float* h;         // [M][H+ctx]
    const float* x;         // [M][3]
    float* hin;             // [M][H]
    float* hres;            // [M][H]
    float* x4;              // [M][4]
    int M, H, ctx;
};

__global__ void k_egcl_node_in(EgclNodeInArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(idx / a.H), k = (int)(idx - (long long)i * a.H);
    if (i >= a.M) return;
    const float v = a.h[(size_t)i * (a.H + a.ctx) + k];
    a.hin[(size_t)i * a.H + k] = v;
    a.hres[(size_t)i * a.H + k] = v;
    if (k < 4) a.x4[(size_t)i * 4 + k] = (k < 3) ? a.x[(size_t)i * 3 + k] : 0.0f;
}

// outputs: h_out = [h_new | context] * node_mask,  x_out = (x + sum of incoming trans) * node_mask      (gcl.py:186-190)
struct EgclNodeOutArgs {
    const float* hnew;      // [M][H]
    const float* hin;       // [M][H] (context columns)
    const float* x4;        // [M][4]
    const float* xagg;      // [M][4] or NULL (no coordinate update)
    const float* nmask;     // [M] or NULL
    float* h_out;           // [M][H+ctx]
    float* x_out;           // [M][3]
    int M, H, ctx;
};

__global__ void k_egcl_node_out(EgclNodeOutArgs a) {
    const int W = a.H + a.ctx;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(idx / W), k = (int)(idx - (long long)i * W);
    if (i >= a.M) return;
    const float m = a.nmask ? a.nmask[i] : 1.0f;
    const float v = (k < a.H) ? a.hnew[(size_t)i * a.H + k] : a.hin[(size_t)i * a.H + a.H - a.ctx + (k - a.H)];
    a.h_out[(size_t)i * W + k] = v * m;
    if (k < 3) a.x_out[(size_t)i * 3 + k] = (a.x4[(size_t)i * 4 + k] + (a.xagg ? a.xagg[(size_t)i * 4 + k] : 0.0f)) * m;
}

// ----------------------------------------------------------------------------- small dense layer (stage-2 embeddings / heads)
// y[m][n] = act(sum_k x[m][k] W[n][k] + b[n]); one thread per output element, one fmaf chain over k.  The layers it serves
// are tiny (edge_denoise.py:29-33, 55-57: K = 2 ... 3H + 1, N = 1 ... vocabulary size, M = beam-sized): launch-bound.
struct LinArgs {
    const float* x; const float* W; const float* b; float* y;
    int M, K, N, ldx, ldy, act;
};

__global__ void k_linear(LinArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.M * a.N) return;
    const int m = (int)(idx / a.N), n = (int)(idx - (long long)m * a.N);
    const float* xr = a.x + (size_t)m * a.ldx;
    const float* wr = a.W + (size_t)n * a.K;
    float acc = 0.f;
    int k = 0;
    if ((a.K & 3) == 0 && (a.ldx & 3) == 0) {
        for (; k < a.K; k += 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + k), wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(xv[j], wv[j], acc);
        }
    }
    for (; k < a.K; ++k) acc = __builtin_fmaf(xr[k], wr[k], acc);
    if (a.b) acc += a.b[n];
    if (a.act == 1) acc = silu_f(acc);
    else if (a.act == 2) acc = sigmoid_f(acc);
    a.y[(size_t)m * a.ldy + n] = acc;
}
