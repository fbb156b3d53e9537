// Host side of libhierdiff_hip.so: the C ABI declared in include/hierdiff_hip.h.
// gfx950 only; built with hipcc --offload-arch=gfx950 (see hierdiff_amd/build.py).
#include "../../include/hierdiff_hip.h"
#include "kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

static int g_edge_grid = 512;      // 2 x CUs (set in hd_create); the experimental persistent k_edge_p launches half of it

// ----------------------------------------------------------------------------- errors

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(HD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

#define HD_TRY(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != HD_OK) return _r;  \
    } while (0)

// ----------------------------------------------------------------------------- objects

struct LayerW {                 // float offsets into hd_handle::dw
    size_t ab_img, ab_bias, wrd, w2_img, b2, wa, w3_img, b3, w4_img, b4;
    float ba;
};

struct ProfRec { int fam; hipEvent_t a, b; };

struct hd_handle {
    hd_config cfg;
    int device;
    int H, fin, F, D, NS;       // NS: 32-column sub-tiles per node-GEMM workgroup tile
    bool fused;                 // bf16x3: one k_node launch per node update instead of k_gemm x3 + k_agg
    long long n_weights;
    bool weights_set;
    float* dw;                  // packed weights
    size_t dw_floats;
    size_t embT, emb_b, outW, out_b;
    std::vector<LayerW> gcl;    // [n_layers * inv_sublayers]
    std::vector<LayerW> coord;  // [n_layers]
    // device scalars
    int* d_nanflag;
    long long* d_nan_events;
    // schedule
    int T;
    std::vector<float> tau_h, coef_h;
    float* d_tau;
    float* d_coef;
    // graph-replay state
    int* d_step;
    uint32_t* d_draw;
    float* d_tcur;
    hipStream_t own_stream;     // capture stream used when the caller passes the legacy NULL stream
    // profiling
    int prof;                   // bitmask of kernel families bracketed with events
    int prof_stride;            // bracket every prof_stride-th forward
    long long prof_fwd;
    bool prof_now;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used;
};

struct hd_topology {
    hd_handle* h;
    int device;
    int B, N, M, M_pad, E, E_pad, n_tiles, n_wg, n_parts;
    // device tables
    int *node_of, *slot_of, *ei, *ej, *tile_pbase, *tile_nseg, *pstart, *nvalid;
    uint8_t *eseg, *nm_bytes;
    float* nmask;
    // workspace
    float *hbuf, *AB, *AB2, *Tb, *agg, *x0, *xcur, *part, *xpart, *eps;
};

// ----------------------------------------------------------------------------- small helpers

extern "C" int hd_version(void) { return HD_ABI_VERSION; }
extern "C" const char* hd_last_error(void) { return g_err.c_str(); }

extern "C" int hd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

static long long weight_count(const hd_config& c) {
    const long long H = c.hidden_nf, fin = c.in_node_nf + c.context_node_nf;
    long long n = H * fin + H + fin * H + fin;
    long long gcl = H * (2 * H + 2) + H + H * H + H + H * 2 * H + H + H * H + H + (c.attention ? H + 1 : 0);
    long long crd = H * (2 * H + 2) + H + H * H + H + H;
    n += (long long)c.n_layers * (c.inv_sublayers * gcl + crd);
    return n;
}

extern "C" long long hd_weight_count(const hd_handle* h) { return h ? h->n_weights : 0; }

template <typename T>
static int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) return fail(HD_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return HD_OK;
}

template <typename T>
static int dev_upload(T** p, const std::vector<T>& v) {
    HD_TRY(dev_alloc(p, v.size()));
    if (!v.empty()) HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return HD_OK;
}

// ----------------------------------------------------------------------------- create / destroy

static int prepare_kernels(int H);

extern "C" int hd_create(const hd_config* cfg, int device, hd_handle** out) {
    if (!cfg || !out) return fail(HD_E_INVALID, "hd_create: null argument");
    *out = nullptr;
    if (cfg->n_dims != 3) return fail(HD_E_INVALID, "hd_create: n_dims must be 3");
    const int H = cfg->hidden_nf;
    if (H != 32 && H != 64 && H != 128 && H != 256)
        return fail(HD_E_INVALID, "hd_create: hidden_nf must be 32, 64, 128 or 256");
    if (cfg->n_layers < 1 || cfg->inv_sublayers < 1) return fail(HD_E_INVALID, "hd_create: n_layers / inv_sublayers must be >= 1");
    const int F = cfg->in_node_nf - (cfg->condition_time ? 1 : 0);
    if (F < 1) return fail(HD_E_INVALID, "hd_create: in_node_nf must leave at least one feature column");
    if (cfg->context_node_nf < 0) return fail(HD_E_INVALID, "hd_create: context_node_nf < 0");
    if (cfg->precision != 0 && cfg->precision != 1) return fail(HD_E_INVALID, "hd_create: precision must be 0 (fp32) or 1 (bf16x3)");
    if (!(cfg->normalization_factor != 0.0f)) return fail(HD_E_INVALID, "hd_create: normalization_factor == 0");
    if (hd_device_count() <= device || device < 0)
        return fail(HD_E_HIP, "hd_create: no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    hd_handle* h = new hd_handle();
    h->cfg = *cfg;
    h->device = device;
    h->H = H;
    h->fin = cfg->in_node_nf + cfg->context_node_nf;
    h->F = F;
    h->D = 3 + F;
    h->NS = (H == 32) ? 1 : 2;
    h->fused = cfg->precision == 1;
    h->n_weights = weight_count(*cfg);
    h->weights_set = false;
    h->dw = nullptr;
    h->dw_floats = 0;
    h->T = 0;
    h->d_tau = h->d_coef = nullptr;
    h->prof = 0;
    h->prof_stride = 1;
    h->prof_fwd = 0;
    h->prof_now = false;
    h->pool_used = 0;
    h->own_stream = nullptr;
    int r = dev_alloc(&h->d_nanflag, 1);
    if (r == HD_OK) r = dev_alloc(&h->d_nan_events, 1);
    if (r == HD_OK) r = dev_alloc(&h->d_step, 1);
    if (r == HD_OK) r = dev_alloc(&h->d_draw, 1);
    if (r == HD_OK) r = dev_alloc(&h->d_tcur, 1);
    if (r != HD_OK) { delete h; return r; }
    HIP_TRY(hipMemset(h->d_nanflag, 0, sizeof(int)));
    HIP_TRY(hipMemset(h->d_nan_events, 0, sizeof(long long)));
    r = prepare_kernels(H);
    if (r != HD_OK) { hd_destroy(h); return r; }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            g_edge_grid = 2 * prop.multiProcessorCount;
        if (const char* e = getenv("HD_EDGE_GRID")) g_edge_grid = std::max(1, atoi(e));
    }
    *out = h;
    return HD_OK;
}

extern "C" int hd_destroy(hd_handle* h) {
    if (!h) return HD_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    hipFree(h->dw); hipFree(h->d_nanflag); hipFree(h->d_nan_events);
    hipFree(h->d_tau); hipFree(h->d_coef); hipFree(h->d_step); hipFree(h->d_draw); hipFree(h->d_tcur);
    for (auto e : h->pool) hipEventDestroy(e);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return HD_OK;
}

// ----------------------------------------------------------------------------- weight packing

// B-operand image of the node GEMM for Wt[k][col] = W(col, k): per (col tile, K chunk) a block of
// [WN][4 q][64 lanes][4 j] floats with k = 32c + 16*(lane>>5) + 4q + j, col = ct*32*WN + 32*wc + (lane&31).
template <typename Fn>
static void pack_gemm_b(std::vector<float>& dst, size_t off, int K, int Nc, int WN, Fn W) {
    const int BN = 32 * WN, ntile = Nc / BN, nchunk = K / 32;
    for (int ct = 0; ct < ntile; ++ct)
        for (int c = 0; c < nchunk; ++c)
            for (int wc = 0; wc < WN; ++wc)
                for (int q = 0; q < 4; ++q)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const int k = 32 * c + 16 * (lane >> 5) + 4 * q + j;
                            const int col = ct * BN + 32 * wc + (lane & 31);
                            dst[off + ((((size_t)(ct * nchunk + c) * WN + wc) * 4 + q) * 64 + lane) * 4 + j] = W(col, k);
                        }
}

// B-operand image of the edge kernel: per K chunk [4 q][H/32 ct][64 lanes][4 j],
// k = 32c + 16*(lane>>5) + 4q + j, col = 32ct + (lane&31), value W2[col][k].
static void pack_edge_w2(std::vector<float>& dst, size_t off, int H, const float* W2) {
    const int NCT = H / 32;
    for (int c = 0; c < H / 32; ++c)
        for (int q = 0; q < 4; ++q)
            for (int ct = 0; ct < NCT; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int k = 32 * c + 16 * (lane >> 5) + 4 * q + j;
                        const int col = 32 * ct + (lane & 31);
                        dst[off + (size_t)c * 32 * H + ((size_t)(q * NCT + ct) * 64 + lane) * 4 + j] = W2[(size_t)col * H + k];
                    }
}

// bf16 round-to-nearest-even of an fp32 value (bit pattern in the low 16 bits)
static inline uint16_t bf16_rne(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}
static inline void bf16_split(float v, uint16_t& hi, uint16_t& lo) {
    hi = bf16_rne(v);
    lo = bf16_rne(v - bf16_to_f32(hi));
}

// bf16x3 images (same byte size as the fp32 ones: 2 B head + 2 B tail per weight).
// fused node kernel (k_node): [k-step s][column tile ct][hi|lo][64 lanes][8], k = 16s + 8*(lane>>5) + i, col = 32ct + (lane&31).
template <typename Fn>
static void pack_node_b(std::vector<float>& dstf, size_t off, int K, int Nc, Fn W) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dstf.data() + off);
    const int nct = Nc / 32;
    for (int st = 0; st < K / 16; ++st)
        for (int ct = 0; ct < nct; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    uint16_t hi, lo;
                    bf16_split(W(32 * ct + (lane & 31), 16 * st + 8 * (lane >> 5) + i), hi, lo);
                    const size_t base = ((size_t)(st * nct + ct) * 2) * 512;
                    dst[base + (size_t)lane * 8 + i] = hi;
                    dst[base + 512 + (size_t)lane * 8 + i] = lo;
                }
}

// edge kernel: per K chunk [hi|lo][2 k-steps][H/32 ct][64 lanes][8], k = 32c + 16*(lane>>5) + 8s + i.
static void pack_edge_w2_bf(std::vector<float>& dstf, size_t off, int H, const float* W2) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dstf.data() + off);
    const int NCT = H / 32;
    for (int c = 0; c < H / 32; ++c)
        for (int st = 0; st < 2; ++st)
            for (int ct = 0; ct < NCT; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int k = 32 * c + 16 * (lane >> 5) + 8 * st + i;
                        const int col = 32 * ct + (lane & 31);
                        uint16_t hi, lo;
                        bf16_split(W2[(size_t)col * H + k], hi, lo);
                        const size_t blk = (size_t)c * 32 * H * 2;
                        dst[blk + (((size_t)(0 * 2 + st) * NCT + ct) * 64 + lane) * 8 + i] = hi;
                        dst[blk + (((size_t)(1 * 2 + st) * NCT + ct) * 64 + lane) * 8 + i] = lo;
                    }
}

extern "C" int hd_set_weights(hd_handle* h, const float* blob, long long n, int on_device, void* stream) {
    if (!h || !blob) return fail(HD_E_INVALID, "hd_set_weights: null argument");
    if (n != h->n_weights)
        return fail(HD_E_INVALID, "hd_set_weights: expected " + std::to_string(h->n_weights) + " values, got " + std::to_string(n));
    HIP_TRY(hipSetDevice(h->device));
    std::vector<float> host;
    const float* src = blob;
    if (on_device) {
        host.resize((size_t)n);
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipMemcpy(host.data(), blob, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        src = host.data();
    }
    const hd_config& c = h->cfg;
    const int H = h->H, fin = h->fin, WN = h->NS;
    const int L = c.n_layers, S = c.inv_sublayers;
    const bool bf = c.precision != 0;
    // layout of the packed buffer
    size_t off = 0;
    auto take = [&](size_t cnt) { size_t o = off; off += (cnt + 3) & ~size_t(3); return o; };
    h->embT = take((size_t)fin * H); h->emb_b = take(H); h->outW = take((size_t)fin * H); h->out_b = take(fin);
    h->gcl.assign((size_t)L * S, LayerW());
    h->coord.assign((size_t)L, LayerW());
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < S; ++j) {
            LayerW& w = h->gcl[(size_t)i * S + j];
            w.ab_img = take((size_t)H * 2 * H); w.ab_bias = take(2 * H); w.wrd = take(2 * H);
            w.w2_img = take((size_t)H * H); w.b2 = take(H); w.wa = take(H);
            w.w3_img = take((size_t)2 * H * H); w.b3 = take(H); w.w4_img = take((size_t)H * H); w.b4 = take(H);
        }
        LayerW& w = h->coord[i];
        w.ab_img = take((size_t)H * 2 * H); w.ab_bias = take(2 * H); w.wrd = take(2 * H);
        w.w2_img = take((size_t)H * H); w.b2 = take(H); w.wa = take(H);
        w.w3_img = w.b3 = w.w4_img = w.b4 = 0;
    }
    std::vector<float> pk(off, 0.0f);
    // walk the canonical blob
    const float* p = src;
    auto next = [&](size_t cnt) { const float* q = p; p += cnt; return q; };
    {
        const float* We = next((size_t)H * fin);      // embedding.weight [H][fin]
        const float* be = next(H);
        const float* Wo = next((size_t)fin * H);      // embedding_out.weight [fin][H]
        const float* bo = next(fin);
        for (int f = 0; f < fin; ++f)
            for (int k = 0; k < H; ++k) pk[h->embT + (size_t)f * H + k] = We[(size_t)k * fin + f];
        std::copy(be, be + H, pk.begin() + h->emb_b);
        std::copy(Wo, Wo + (size_t)fin * H, pk.begin() + h->outW);
        std::copy(bo, bo + fin, pk.begin() + h->out_b);
    }
    // bf16x3 mode runs the edge model in a scaled domain (see silu_scaled in common.hpp): everything feeding a
    // SiLU / sigmoid of the edge kernel carries c = -log2(e), its consumers carry 1/c.  One rounding per weight.
    const double cs = bf ? -1.4426950408889634074 : 1.0, cs_inv = 1.0 / cs;
    auto sc = [&](float v) { return (float)((double)v * cs); };
    auto sc_inv = [&](float v) { return (float)((double)v * cs_inv); };
    auto pack_first = [&](LayerW& w, const float* W1, const float* b1) {
        // W1 [H][2H+2]: columns [h_row(H) | h_col(H) | radial_cur | radial_init] (egnn_new.py:39,93,144)
        const int ld = 2 * H + 2;
        auto wab = [&](int col, int k) {
            return sc((col < H) ? W1[(size_t)col * ld + k] : W1[(size_t)(col - H) * ld + H + k]);
        };
        if (h->fused) pack_node_b(pk, w.ab_img, H, 2 * H, wab);
        else pack_gemm_b(pk, w.ab_img, H, 2 * H, WN, wab);
        for (int k = 0; k < H; ++k) {
            pk[w.ab_bias + k] = sc(b1[k]);
            pk[w.ab_bias + H + k] = 0.0f;
            pk[w.wrd + k] = sc(W1[(size_t)k * ld + 2 * H]);
            pk[w.wrd + H + k] = sc(W1[(size_t)k * ld + 2 * H + 1]);
        }
    };
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < S; ++j) {
            LayerW& w = h->gcl[(size_t)i * S + j];
            const float* W1 = next((size_t)H * (2 * H + 2)); const float* b1 = next(H);
            const float* W2 = next((size_t)H * H);           const float* b2 = next(H);
            const float* W3 = next((size_t)H * 2 * H);       const float* b3 = next(H);
            const float* W4 = next((size_t)H * H);           const float* b4 = next(H);
            pack_first(w, W1, b1);
            // node_mlp.0: columns k >= H multiply the neighbour sums, which arrive scaled by c
            auto w3 = [&](int col, int k) { const float v = W3[(size_t)col * 2 * H + k]; return k >= H ? sc_inv(v) : v; };
            auto w4 = [&](int col, int k) { return W4[(size_t)col * H + k]; };
            if (bf) pack_edge_w2_bf(pk, w.w2_img, H, W2);
            else pack_edge_w2(pk, w.w2_img, H, W2);
            if (h->fused) {
                pack_node_b(pk, w.w3_img, 2 * H, H, w3);
                pack_node_b(pk, w.w4_img, H, H, w4);
            } else {
                pack_gemm_b(pk, w.w3_img, 2 * H, H, WN, w3);
                pack_gemm_b(pk, w.w4_img, H, H, WN, w4);
            }
            for (int k = 0; k < H; ++k) pk[w.b2 + k] = sc(b2[k]);
            std::copy(b3, b3 + H, pk.begin() + w.b3);
            std::copy(b4, b4 + H, pk.begin() + w.b4);
            if (c.attention) {
                const float* wa = next(H); const float* ba = next(1);
                std::copy(wa, wa + H, pk.begin() + w.wa);
                w.ba = sc(ba[0]);
            } else {
                w.ba = 0.0f;
            }
        }
        LayerW& w = h->coord[i];
        const float* W5 = next((size_t)H * (2 * H + 2)); const float* b5 = next(H);
        const float* W6 = next((size_t)H * H);           const float* b6 = next(H);
        const float* w7 = next(H);
        pack_first(w, W5, b5);
        if (bf) pack_edge_w2_bf(pk, w.w2_img, H, W6);
        else pack_edge_w2(pk, w.w2_img, H, W6);
        for (int k = 0; k < H; ++k) { pk[w.b2 + k] = sc(b6[k]); pk[w.wa + k] = sc_inv(w7[k]); }
        w.ba = 0.0f;
    }
    if (p - src != n) return fail(HD_E_INVALID, "hd_set_weights: internal layout mismatch");
    if (h->dw_floats != pk.size()) {
        hipFree(h->dw);
        h->dw = nullptr;
        HD_TRY(dev_alloc(&h->dw, pk.size()));
        h->dw_floats = pk.size();
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(h->dw, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    h->weights_set = true;
    return HD_OK;
}

// ----------------------------------------------------------------------------- topology

extern "C" int hd_topology_destroy(hd_topology* t) {
    if (!t) return HD_OK;
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    hipFree(t->node_of); hipFree(t->slot_of); hipFree(t->ei); hipFree(t->ej); hipFree(t->tile_pbase);
    hipFree(t->tile_nseg); hipFree(t->pstart); hipFree(t->nvalid); hipFree(t->eseg); hipFree(t->nm_bytes);
    hipFree(t->nmask); hipFree(t->hbuf); hipFree(t->AB); hipFree(t->AB2); hipFree(t->Tb); hipFree(t->agg); hipFree(t->x0); hipFree(t->xcur);
    hipFree(t->part); hipFree(t->xpart); hipFree(t->eps);
    delete t;
    return HD_OK;
}

extern "C" int hd_topology_create(hd_handle* h, const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N,
                                  hd_topology** out) {
    if (!h || !node_mask || !out) return fail(HD_E_INVALID, "hd_topology_create: null argument");
    *out = nullptr;
    if (B < 1 || N < 1) return fail(HD_E_INVALID, "hd_topology_create: B and N must be >= 1");
    if ((long long)B * N > (1LL << 30)) return fail(HD_E_INVALID, "hd_topology_create: B*N too large");
    HIP_TRY(hipSetDevice(h->device));
    const size_t BN = (size_t)B * N;
    // active nodes: masked-in, or touched by an unmasked edge (general edge masks only)
    std::vector<uint8_t> active(node_mask, node_mask + BN);
    for (auto& a : active) a = a ? 1 : 0;
    if (edge_mask) {
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j)
                    if (edge_mask[((size_t)b * N + i) * N + j]) { active[(size_t)b * N + i] = 1; active[(size_t)b * N + j] = 1; }
    }
    std::vector<int> slot_of(BN, -1), node_of, nvalid(B, 0);
    std::vector<float> nmask;
    for (size_t f = 0; f < BN; ++f) {
        if (active[f]) { slot_of[f] = (int)node_of.size(); node_of.push_back((int)f); nmask.push_back(node_mask[f] ? 1.0f : 0.0f); }
        if (node_mask[f]) nvalid[f / N]++;
    }
    const int M = (int)node_of.size();
    const int M_pad = std::max(128, (M + 127) / 128 * 128);
    nmask.resize(M_pad, 0.0f);
    std::vector<int> ei, ej;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const size_t fi = (size_t)b * N + i;
            if (!active[fi]) continue;
            for (int j = 0; j < N; ++j) {
                const size_t fj = (size_t)b * N + j;
                const bool on = edge_mask ? edge_mask[fi * N + j] != 0 : (node_mask[fi] && node_mask[fj] && i != j);
                if (on) { ei.push_back(slot_of[fi]); ej.push_back(slot_of[fj]); }
            }
        }
    const long long E = (long long)ei.size();
    if (E > (1LL << 30)) return fail(HD_E_INVALID, "hd_topology_create: too many edges");
    const int E_pad = (int)((E + 127) / 128 * 128);
    const int n_tiles = (int)((E + 31) / 32);
    const int n_wg = E_pad / 128;
    ei.resize(E_pad, 0); ej.resize(E_pad, 0);
    std::vector<uint8_t> eseg(E_pad, 255);
    std::vector<int> tile_pbase(std::max(1, E_pad / 32), 0), tile_nseg(std::max(1, E_pad / 32), 0);
    std::vector<int> part_node;
    for (int t = 0; t < E_pad / 32; ++t) {
        tile_pbase[t] = (int)part_node.size();
        int prev = -1, seg = -1;
        for (int r = 0; r < 32; ++r) {
            const long long e = (long long)t * 32 + r;
            if (e >= E) break;
            if (ei[e] != prev) { ++seg; prev = ei[e]; part_node.push_back(prev); }
            eseg[e] = (uint8_t)seg;
        }
        tile_nseg[t] = seg + 1;
    }
    const int n_parts = (int)part_node.size();
    std::vector<int> pstart(M + 1, 0);
    for (int p = 0; p < n_parts; ++p) pstart[part_node[p] + 1]++;
    for (int i = 0; i < M; ++i) pstart[i + 1] += pstart[i];
    std::vector<uint8_t> nm_bytes(BN);
    for (size_t f = 0; f < BN; ++f) nm_bytes[f] = node_mask[f] ? 1 : 0;

    hd_topology* t = new hd_topology();
    std::memset(t, 0, sizeof(*t));
    t->h = h; t->device = h->device; t->B = B; t->N = N; t->M = M; t->M_pad = M_pad; t->E = (int)E; t->E_pad = E_pad;
    t->n_tiles = n_tiles; t->n_wg = n_wg; t->n_parts = n_parts;
    const int H = h->H;
    int r = HD_OK;
    auto ok = [&](int rc) { if (r == HD_OK) r = rc; };
    ok(dev_upload(&t->node_of, node_of)); ok(dev_upload(&t->slot_of, slot_of)); ok(dev_upload(&t->ei, ei));
    ok(dev_upload(&t->ej, ej)); ok(dev_upload(&t->tile_pbase, tile_pbase)); ok(dev_upload(&t->tile_nseg, tile_nseg));
    ok(dev_upload(&t->pstart, pstart)); ok(dev_upload(&t->nvalid, nvalid)); ok(dev_upload(&t->eseg, eseg));
    ok(dev_upload(&t->nm_bytes, nm_bytes)); ok(dev_upload(&t->nmask, nmask));
    ok(dev_alloc(&t->hbuf, (size_t)M_pad * H)); ok(dev_alloc(&t->AB, (size_t)M_pad * 2 * H));
    ok(dev_alloc(&t->AB2, h->fused ? (size_t)M_pad * 2 * H : 1));
    ok(dev_alloc(&t->Tb, (size_t)M_pad * H)); ok(dev_alloc(&t->agg, (size_t)M_pad * H)); ok(dev_alloc(&t->x0, (size_t)M_pad * 4));
    ok(dev_alloc(&t->xcur, (size_t)M_pad * 4)); ok(dev_alloc(&t->part, (size_t)std::max(1, n_parts) * H));
    ok(dev_alloc(&t->xpart, (size_t)std::max(1, n_parts) * 4)); ok(dev_alloc(&t->eps, BN * h->D));
    if (r != HD_OK) { hd_topology_destroy(t); return r; }
    // pad rows stay zero for the lifetime of the topology (kernels never write them)
    hipMemset(t->hbuf, 0, (size_t)M_pad * H * 4); hipMemset(t->AB, 0, (size_t)M_pad * 2 * H * 4);
    if (h->fused) hipMemset(t->AB2, 0, (size_t)M_pad * 2 * H * 4);
    hipMemset(t->Tb, 0, (size_t)M_pad * H * 4); hipMemset(t->agg, 0, (size_t)M_pad * H * 4); hipMemset(t->x0, 0, (size_t)M_pad * 16);
    hipMemset(t->xcur, 0, (size_t)M_pad * 16);
    HIP_TRY(hipDeviceSynchronize());
    *out = t;
    return HD_OK;
}

extern "C" int hd_topology_info(const hd_topology* t, long long* info6) {
    if (!t || !info6) return fail(HD_E_INVALID, "hd_topology_info: null argument");
    info6[0] = t->B; info6[1] = t->N; info6[2] = t->M; info6[3] = t->E; info6[4] = t->n_tiles; info6[5] = t->n_parts;
    return HD_OK;
}

// ----------------------------------------------------------------------------- profiling

extern "C" int hd_profile_enable(hd_handle* h, int on) {
    if (!h) return fail(HD_E_INVALID, "hd_profile_enable: null handle");
    h->prof = on & 7;
    h->prof_stride = std::max(1, on >> 8);
    h->prof_fwd = 0;
    h->prof_now = false;
    h->recs.clear();
    h->pool_used = 0;
    return HD_OK;
}

static hipEvent_t prof_event(hd_handle* h) {
    if (h->pool_used == h->pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        h->pool.push_back(e);
    }
    return h->pool[h->pool_used++];
}

struct ProfScope {
    hd_handle* h; hipStream_t s; int fam; hipEvent_t a;
    ProfScope(hd_handle* h_, hipStream_t s_, int fam_) : h(h_), s(s_), fam(fam_), a(nullptr) {
        if (h->prof_now && (h->prof & (1 << fam))) { a = prof_event(h); hipEventRecord(a, s); }
    }
    ~ProfScope() {
        if (a) { hipEvent_t b = prof_event(h); hipEventRecord(b, s); h->recs.push_back({fam, a, b}); }
    }
};

extern "C" int hd_profile_read(hd_handle* h, double* ms3, long long* launches3) {
    if (!h || !ms3 || !launches3) return fail(HD_E_INVALID, "hd_profile_read: null argument");
    for (int k = 0; k < 3; ++k) { ms3[k] = 0.0; launches3[k] = 0; }
    for (auto& r : h->recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        ms3[r.fam] += ms;
        launches3[r.fam] += 1;
    }
    h->recs.clear();
    h->pool_used = 0;
    return HD_OK;
}

// ----------------------------------------------------------------------------- forward

template <int WM, int WN, int CN>
static void launch_gemm(int epi, bool cat, const GemmArgs& g, hipStream_t s) {
    const int nrt = (g.M + 32 * WM - 1) / (32 * WM), nct = g.Nc / (32 * WN * CN);
    dim3 grid(8 * ((nrt + 7) / 8) * nct);
    dim3 block(WM * WN * 64);
    if (cat) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_BIAS_SILU, true>), grid, block, 0, s, g);
    else if (epi == EPI_BIAS) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_BIAS, false>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_RESID_MASK, false>), grid, block, 0, s, g);
}

// Node-GEMM tile shape per hidden size: (waves M, waves N, accumulators per wave); the weight images are
// packed for the matching number of 32-column sub-tiles NS = WN*CN.
static void gemm(hd_handle* h, int epi, bool cat, const GemmArgs& g, hipStream_t s) {
    ProfScope ps(h, s, 1);

    if (h->NS == 1) launch_gemm<4, 1, 1>(epi, cat, g, s);        // H = 32: 128 x 32 tiles
    else launch_gemm<2, 2, 1>(epi, cat, g, s);                   // 64 x 64 tiles (fastest measured)
}

// Fused node update (bf16x3 only): min(4, H/32) wavefronts per 32-row workgroup.
template <int H>
static int node_lds_bytes(bool upd) { return 32 * ((upd ? 2 * H : H) + 8) * 4 + 32 * (H + 8) * 4; }

template <int H, int NW>
static void launch_node_hw(bool upd, int nab, const NodeArgs& a, hipStream_t s) {
    const int nrt = (a.M + 31) / 32;
    const dim3 grid(8 * ((nrt + 7) / 8)), block(64 * NW);
    const int lds = node_lds_bytes<H>(upd);
    if (!upd) hipLaunchKernelGGL((k_node<H, NW, false, 1>), grid, block, lds, s, a);
    else if (nab == 1) hipLaunchKernelGGL((k_node<H, NW, true, 1>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((k_node<H, NW, true, 2>), grid, block, lds, s, a);
}

template <int H, int NW>
static int prepare_node_hw() {
    HIP_TRY(hipFuncSetAttribute((const void*)k_node<H, NW, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, node_lds_bytes<H>(false)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_node<H, NW, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, node_lds_bytes<H>(true)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_node<H, NW, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, node_lds_bytes<H>(true)));
    return HD_OK;
}

// wavefronts per 32-row workgroup: one 32-column tile of an H-wide output each, at most 8 (two per SIMD)
template <int H>
static void launch_node_h(bool upd, int nab, const NodeArgs& a, hipStream_t s) {
    launch_node_hw<H, (H / 32 < 8 ? H / 32 : 8)>(upd, nab, a, s);
}

template <int H>
static int prepare_node_h() { return prepare_node_hw<H, (H / 32 < 8 ? H / 32 : 8)>(); }

static void node_update(hd_handle* h, bool upd, int nab, const NodeArgs& a, hipStream_t s) {
    ProfScope ps(h, s, 1);
    switch (h->H) {
        case 32: launch_node_h<32>(upd, nab, a, s); break;
        case 64: launch_node_h<64>(upd, nab, a, s); break;
        case 128: launch_node_h<128>(upd, nab, a, s); break;
        default: launch_node_h<256>(upd, nab, a, s); break;
    }
}

template <int H>
static int edge_p_lds_bytes() { return (3 * 32 * H + 4 * 2048 + 4 * 144) * 4; }    // k_edge_p: three chunk buffers, AB row slots, per-wave scratch

template <int H>
static int edge_lds_bytes() { return (2 * 32 * H + 2 * H + 4 * 136) * 4; }   // dynamic part (w_r/w_d/b2/wa are static)

static long long* g_trace = nullptr;
static int g_trace_wg = 0;
// debug only (HD_ABLATE=16): cycle stamps of the last traced edge launch, 32 values per workgroup
extern "C" int hd_debug_edge_trace(long long* out, int max_wg) {
    if (!g_trace) return 0;
    const int n = std::min(max_wg, g_trace_wg);
    hipDeviceSynchronize();
    hipMemcpy(out, g_trace, sizeof(long long) * 32 * n, hipMemcpyDeviceToHost);
    return n;
}

template <int H>
static int launch_edge_h(int prec, bool coord, const EdgeArgs& a, hipStream_t s) {
    const int lds = edge_lds_bytes<H>();
    const dim3 grid(a.n_wg), block(256);
    if constexpr (H == 256) {
        static int abl = -1;
        if (abl < 0) { const char* e = getenv("HD_ABLATE"); abl = e ? atoi(e) : 0; }
        if (abl && prec == 1 && !coord) {
            auto run = [&](auto Abl) {
                constexpr int ABL = decltype(Abl)::value;
                static bool attr = false;
                if (!attr) {
                    hipFuncSetAttribute((const void*)k_edge<256, false, 1, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                    attr = true;
                }
                EdgeArgs b = a;
                if constexpr (ABL & 16) {
                    if (!g_trace) hipMalloc(reinterpret_cast<void**>(&g_trace), sizeof(long long) * 32 * 4096);
                    if (a.n_wg > 4096) return;
                    b.trace = g_trace; g_trace_wg = a.n_wg;
                }
                hipLaunchKernelGGL((k_edge<256, false, 1, ABL>), grid, block, lds, s, b);
            };
            switch (abl) {
                case 1: run(std::integral_constant<int, 1>{}); return HD_OK;
                case 2: run(std::integral_constant<int, 2>{}); return HD_OK;
                case 4: run(std::integral_constant<int, 4>{}); return HD_OK;
                case 8: run(std::integral_constant<int, 8>{}); return HD_OK;
                case 15: run(std::integral_constant<int, 15>{}); return HD_OK;
                case 16: run(std::integral_constant<int, 16>{}); return HD_OK;
                case 18: run(std::integral_constant<int, 18>{}); return HD_OK;
                case 20: run(std::integral_constant<int, 20>{}); return HD_OK;
                case 24: run(std::integral_constant<int, 24>{}); return HD_OK;
                case 30: run(std::integral_constant<int, 30>{}); return HD_OK;
                default: break;
            }
        }
    }
    if constexpr (H >= 256) {
        // Experimental (HD_EDGE_PIPE=1): persistent one-wave-per-SIMD kernel with the previous tile's epilogue folded
        // into the MFMA loop.  Correct (passes the whole GPU suite) but 124 us vs 103 us for k_edge on MI355X: a single
        // wavefront issues at most one instruction per ~4 cycles and this stream needs ~10 per MFMA.
        static int pipe = -1;
        if (pipe < 0) { const char* e = getenv("HD_EDGE_PIPE"); pipe = e ? atoi(e) : 0; }
        if (prec == 1 && pipe) {
            const dim3 pgrid(std::min(a.n_wg, std::max(1, g_edge_grid / 2)));
            const int plds = edge_p_lds_bytes<H>();
            static int ptrace = -1;
            if (ptrace < 0) { const char* e = getenv("HD_EDGE_PTRACE"); ptrace = e ? atoi(e) : 0; }
            if (ptrace && !coord) {
                static bool attr = false;
                if (!attr) { hipFuncSetAttribute((const void*)k_edge_p<H, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, plds); attr = true; }
                EdgeArgs b = a;
                if (!g_trace) hipMalloc(reinterpret_cast<void**>(&g_trace), sizeof(long long) * 32 * 4096);
                b.trace = g_trace; g_trace_wg = (int)pgrid.x;
                hipLaunchKernelGGL((k_edge_p<H, false, true>), pgrid, block, plds, s, b);
                return HD_OK;
            }
            if (coord) hipLaunchKernelGGL((k_edge_p<H, true>), pgrid, block, plds, s, a);
            else hipLaunchKernelGGL((k_edge_p<H, false>), pgrid, block, plds, s, a);
            return HD_OK;
        }
    }
    if (prec == 0) {
        if (coord) hipLaunchKernelGGL((k_edge<H, true, 0>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_edge<H, false, 0>), grid, block, lds, s, a);
    } else {
        if (coord) hipLaunchKernelGGL((k_edge<H, true, 1>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_edge<H, false, 1>), grid, block, lds, s, a);
    }
    return HD_OK;
}

// Raise the dynamic-LDS limit of both edge kernels for this device (done once, at hd_create: it is
// not allowed while a stream is capturing).
template <int H>
static int prepare_edge_h() {
    const int lds = edge_lds_bytes<H>();
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if constexpr (H >= 256) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_p<H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, edge_p_lds_bytes<H>()));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_p<H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, edge_p_lds_bytes<H>()));
    }
    return HD_OK;
}

static int prepare_kernels(int H) {
    switch (H) {
        case 32: HD_TRY(prepare_node_h<32>()); return prepare_edge_h<32>();
        case 64: HD_TRY(prepare_node_h<64>()); return prepare_edge_h<64>();
        case 128: HD_TRY(prepare_node_h<128>()); return prepare_edge_h<128>();
        default: HD_TRY(prepare_node_h<256>()); return prepare_edge_h<256>();
    }
}

static int edge(hd_handle* h, bool coord, const EdgeArgs& a, hipStream_t s) {
    if (a.n_wg == 0) return HD_OK;
    ProfScope ps(h, s, 0);
    switch (h->H) {
        case 32: return launch_edge_h<32>(h->cfg.precision, coord, a, s);
        case 64: return launch_edge_h<64>(h->cfg.precision, coord, a, s);
        case 128: return launch_edge_h<128>(h->cfg.precision, coord, a, s);
        default: return launch_edge_h<256>(h->cfg.precision, coord, a, s);
    }
}

static int forward_impl(hd_handle* h, hd_topology* t, const float* xh, const float* tt, int t_numel,
                        const float* context, int mol_shape, float* out, hipStream_t s) {
    const hd_config& c = h->cfg;
    const int H = h->H, M = t->M;
    const float* W = h->dw;
    h->prof_now = h->prof != 0 && (h->prof_fwd++ % h->prof_stride) == 0;
    if (M == 0) HIP_TRY(hipMemsetAsync(h->d_nanflag, 0, sizeof(int), s));      // otherwise k_node_init resets it
    if (M > 0) {
        {
            ProfScope ps(h, s, 2);
            InitArgs a;
            a.xh = xh; a.t = tt; a.ctx = context; a.node_of = t->node_of; a.nmask = t->nmask;
            a.embT = W + h->embT; a.emb_b = W + h->emb_b; a.h = t->hbuf; a.x0 = t->x0; a.xcur = t->xcur;
            a.nanflag = h->d_nanflag;
            a.M = M; a.N = t->N; a.D = h->D; a.F = h->F; a.C = c.context_node_nf; a.H = H;
            a.t_stride = (t_numel == 1) ? 0 : 1; a.cond_time = c.condition_time;
            const long long total = (long long)M * (H / 4);
            hipLaunchKernelGGL(k_node_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        }
        const float range = c.coords_range / (float)c.n_layers;
        const int S = c.inv_sublayers;
        // fused path: AB of the layer about to run is produced by the previous node update (or, for the very
        // first layer, by an AB-only launch); `ab_cur` is the buffer the next edge kernel reads
        const float* ab_cur = t->AB;
        auto node_args = [&]() {
            NodeArgs a;
            std::memset(&a, 0, sizeof(a));
            a.h_in = t->hbuf; a.h_out = t->hbuf; a.part = t->part; a.pstart = t->pstart; a.nmask = t->nmask;
            a.norm = c.normalization_factor; a.M = M;
            return a;
        };
        auto set_ab = [&](NodeArgs& a, int q, const LayerW& nw, float* dst) {
            a.ABimg[q] = W + nw.ab_img; a.ABbias[q] = W + nw.ab_bias; a.ABout[q] = dst;
        };
        if (h->fused) {
            NodeArgs a = node_args();
            set_ab(a, 0, h->gcl[0], t->AB);
            node_update(h, false, 1, a, s);
        }
        for (int i = 0; i < c.n_layers; ++i) {
            for (int j = 0; j <= c.inv_sublayers; ++j) {
                const bool coord = (j == c.inv_sublayers);
                const LayerW& w = coord ? h->coord[i] : h->gcl[(size_t)i * c.inv_sublayers + j];
                if (!h->fused) {
                    GemmArgs g;
                    std::memset(&g, 0, sizeof(g));
                    g.A = t->hbuf; g.lda = H; g.K1 = H; g.K = H; g.Bimg = W + w.ab_img; g.bias = W + w.ab_bias;
                    g.C = t->AB; g.ldc = 2 * H; g.M = M; g.Nc = 2 * H; g.nmask = t->nmask;
                    gemm(h, EPI_BIAS, false, g, s);
                }
                EdgeArgs e;
                std::memset(&e, 0, sizeof(e));
                e.AB = ab_cur; e.wrd = W + w.wrd; e.W2img = W + w.w2_img; e.b2 = W + w.b2; e.wa = W + w.wa;
                e.ei = t->ei; e.ej = t->ej; e.eseg = t->eseg; e.tile_pbase = t->tile_pbase; e.tile_nseg = t->tile_nseg;
                e.xcur = t->xcur; e.x0 = t->x0; e.part = coord ? t->xpart : t->part; e.ba = w.ba;
                e.norm_constant = c.norm_constant; e.coords_range = range; e.attention = c.attention;
                e.use_tanh = c.tanh; e.n_tiles = t->n_tiles; e.n_wg = t->n_wg;
                HD_TRY(edge(h, coord, e, s));
                if (!coord && h->fused) {
                    NodeArgs a = node_args();
                    a.W3img = W + w.w3_img; a.b3 = W + w.b3; a.W4img = W + w.w4_img; a.b4 = W + w.b4;
                    int nab = 1;
                    if (j + 1 < S) {
                        set_ab(a, 0, h->gcl[(size_t)i * S + j + 1], t->AB);
                    } else {
                        set_ab(a, 0, h->coord[i], t->AB);
                        if (i + 1 < c.n_layers) { set_ab(a, 1, h->gcl[(size_t)(i + 1) * S], t->AB2); nab = 2; }
                    }
                    node_update(h, true, nab, a, s);
                    ab_cur = t->AB;
                } else if (!coord) {
                    GemmArgs g1;
                    std::memset(&g1, 0, sizeof(g1));
                    {
                        ProfScope ps(h, s, 2);
                        AggArgs ag;
                        ag.part = t->part; ag.pstart = t->pstart; ag.agg = t->agg; ag.norm = c.normalization_factor;
                        ag.M = M; ag.H = H;
                        const long long total = (long long)M * (H / 4);
                        hipLaunchKernelGGL(k_agg, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ag);
                    }
                    g1.A = t->hbuf; g1.lda = H; g1.K1 = H; g1.K = 2 * H; g1.A2 = t->agg;
                    g1.Bimg = W + w.w3_img; g1.bias = W + w.b3; g1.C = t->Tb;
                    g1.ldc = H; g1.M = M; g1.Nc = H; g1.nmask = t->nmask;
                    gemm(h, EPI_BIAS_SILU, true, g1, s);
                    GemmArgs g2;
                    std::memset(&g2, 0, sizeof(g2));
                    g2.A = t->Tb; g2.lda = H; g2.K1 = H; g2.K = H; g2.Bimg = W + w.w4_img; g2.bias = W + w.b4;
                    g2.C = t->hbuf; g2.ldc = H; g2.M = M; g2.Nc = H; g2.nmask = t->nmask;
                    gemm(h, EPI_RESID_MASK, false, g2, s);
                } else {
                    ProfScope ps(h, s, 2);
                    XupdArgs x;
                    x.part = t->xpart; x.pstart = t->pstart; x.nmask = t->nmask; x.xcur = t->xcur;
                    x.norm = c.normalization_factor; x.M = M;
                    hipLaunchKernelGGL(k_xupd, dim3((M + 255) / 256), dim3(256), 0, s, x);
                    if (h->fused) ab_cur = t->AB2;          // next block's first GCL (written by the last node update)
                }
            }
        }
        {
            ProfScope ps(h, s, 2);
            Post1Args p;
            p.h = t->hbuf; p.outW = W + h->outW; p.out_b = W + h->out_b; p.x0 = t->x0; p.xcur = t->xcur;
            p.node_of = t->node_of; p.nmask = t->nmask; p.out = out; p.nanflag = h->d_nanflag;
            p.M = M; p.N = t->N; p.D = h->D; p.F = h->F; p.H = H; p.mol_shape = mol_shape;
            hipLaunchKernelGGL(k_post1, dim3((M + 3) / 4), dim3(256), 0, s, p);
        }
    }
    {
        ProfScope ps(h, s, 2);
        Post2Args p;
        p.slot_of = t->slot_of; p.nmask = t->nmask; p.nvalid = t->nvalid; p.out = out; p.nanflag = h->d_nanflag;
        p.nan_events = h->d_nan_events; p.B = t->B; p.N = t->N; p.D = h->D;
        hipLaunchKernelGGL(k_post2, dim3((t->B + 3) / 4), dim3(256), 0, s, p);
    }
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

static int check_ready(hd_handle* h, hd_topology* t, const char* who) {
    if (!h || !t) return fail(HD_E_INVALID, std::string(who) + ": null handle/topology");
    if (t->h != h) return fail(HD_E_INVALID, std::string(who) + ": topology belongs to another handle");
    if (!h->weights_set) return fail(HD_E_STATE, std::string(who) + ": weights not set (hd_set_weights)");
    return HD_OK;
}

extern "C" int hd_egnn_forward(hd_handle* h, hd_topology* topo, const float* xh, const float* t, int t_numel,
                               const float* context, int mol_shape, float* out, void* stream) {
    HD_TRY(check_ready(h, topo, "hd_egnn_forward"));
    if (!xh || !out) return fail(HD_E_INVALID, "hd_egnn_forward: null tensor");
    if (h->cfg.condition_time && (!t || (t_numel != 1 && t_numel != topo->B)))
        return fail(HD_E_INVALID, "hd_egnn_forward: t must have 1 or B elements");
    if (h->cfg.context_node_nf > 0 && !context) return fail(HD_E_INVALID, "hd_egnn_forward: context required");
    if (mol_shape > topo->N) mol_shape = topo->N;
    HIP_TRY(hipSetDevice(h->device));
    return forward_impl(h, topo, xh, t, t_numel, context, mol_shape, out, (hipStream_t)stream);
}

extern "C" int hd_nan_events(hd_handle* h, void* stream, long long* count) {
    if (!h || !count) return fail(HD_E_INVALID, "hd_nan_events: null argument");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(count, h->d_nan_events, sizeof(long long), hipMemcpyDeviceToHost));
    return HD_OK;
}

// ----------------------------------------------------------------------------- sampling maths

static NoiseSrc make_noise(const float* raw_x, const float* raw_h, int rows, uint64_t seed, uint64_t base,
                           uint32_t draw, int share) {
    NoiseSrc n;
    n.raw_x = raw_x; n.raw_h = raw_h; n.rows = rows; n.seed = seed; n.sample_base = base; n.draw = draw; n.share = share;
    return n;
}

static int step_impl(hd_handle* h, hd_topology* t, const float* zt, const float* eps, const float* coef, int coef_rows,
                     const NoiseSrc& ns, int mol, float* zs, int out_stride, const int* step_ptr,
                     const uint32_t* draw_ptr, uint32_t draw0, hipStream_t s) {
    ProfScope ps(h, s, 2);
    StepArgs a;
    a.zt = zt; a.eps = eps; a.coef = coef; a.nm = t->nm_bytes; a.zs = zs; a.noise = ns; a.draw_ptr = draw_ptr;
    a.step_ptr = step_ptr; a.draw0 = draw0; a.coef_rows = coef_rows; a.B = t->B; a.N = t->N; a.D = h->D; a.F = h->F;
    a.mol = mol; a.out_stride = out_stride;
    hipLaunchKernelGGL(k_post_step, dim3(t->B), dim3(256), (size_t)a.mol * a.D * sizeof(float), s, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_posterior_step(hd_handle* h, hd_topology* topo, const float* zt, const float* eps, const float* coef,
                                 int coef_rows, const float* raw_x, const float* raw_h, int noise_rows, int mol_shape,
                                 float* zs, void* stream) {
    if (!h || !topo) return fail(HD_E_INVALID, "hd_posterior_step: null handle/topology");
    if (!zt || !eps || !coef || !raw_x || !raw_h || !zs) return fail(HD_E_INVALID, "hd_posterior_step: null tensor");
    if (coef_rows != 1 && coef_rows != topo->B) return fail(HD_E_INVALID, "hd_posterior_step: coef_rows must be 1 or B");
    if (noise_rows != 1 && noise_rows != topo->B) return fail(HD_E_INVALID, "hd_posterior_step: noise_rows must be 1 or B");
    const int mol = (mol_shape < 0 || mol_shape > topo->N) ? topo->N : mol_shape;
    if (zs == zt && mol != topo->N) return fail(HD_E_INVALID, "hd_posterior_step: in-place needs mol_shape == N");
    HIP_TRY(hipSetDevice(h->device));
    return step_impl(h, topo, zt, eps, coef, coef_rows, make_noise(raw_x, raw_h, noise_rows, 0, 0, 0, 0), mol, zs, mol,
                     nullptr, nullptr, 0, (hipStream_t)stream);
}

extern "C" int hd_final_decode(hd_handle* h, hd_topology* topo, const float* z0, const float* eps, const float* coef3,
                               const float* raw_x, const float* raw_h, int noise_rows, uint64_t seed,
                               uint64_t sample_id_base, uint32_t draw, int share_rows, float* x, float* hfeat,
                               void* stream) {
    if (!h || !topo) return fail(HD_E_INVALID, "hd_final_decode: null handle/topology");
    if (!z0 || !eps || !coef3 || !x || !hfeat) return fail(HD_E_INVALID, "hd_final_decode: null tensor");
    if ((raw_x == nullptr) != (raw_h == nullptr)) return fail(HD_E_INVALID, "hd_final_decode: raw_x and raw_h go together");
    if (raw_x && noise_rows != 1 && noise_rows != topo->B) return fail(HD_E_INVALID, "hd_final_decode: noise_rows must be 1 or B");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(h, s, 2);
    DecodeArgs a;
    a.z0 = z0; a.eps = eps; a.nm = topo->nm_bytes; a.x = x; a.hfeat = hfeat;
    a.noise = make_noise(raw_x, raw_h, noise_rows, seed, sample_id_base, draw, share_rows);
    a.sigma_0 = coef3[0]; a.alpha_0 = coef3[1]; a.sigma_x = coef3[2];
    a.B = topo->B; a.N = topo->N; a.D = h->D; a.F = h->F;
    hipLaunchKernelGGL(k_final_decode, dim3((topo->B + 3) / 4), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_noise(hd_handle* h, hd_topology* topo, const float* raw_x, const float* raw_h, int noise_rows,
                        uint64_t seed, uint64_t sample_id_base, uint32_t draw, int share_rows, float* z, void* stream) {
    if (!h || !topo || !z) return fail(HD_E_INVALID, "hd_noise: null argument");
    if ((raw_x == nullptr) != (raw_h == nullptr)) return fail(HD_E_INVALID, "hd_noise: raw_x and raw_h go together");
    if (raw_x && noise_rows != 1 && noise_rows != topo->B) return fail(HD_E_INVALID, "hd_noise: noise_rows must be 1 or B");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(h, s, 2);
    NoiseArgs a;
    a.nm = topo->nm_bytes; a.z = z; a.noise = make_noise(raw_x, raw_h, noise_rows, seed, sample_id_base, draw, share_rows);
    a.B = topo->B; a.N = topo->N; a.D = h->D; a.F = h->F;
    hipLaunchKernelGGL(k_noise, dim3((topo->B + 3) / 4), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_set_schedule(hd_handle* h, int T, const float* tau, const float* coef4) {
    if (!h || !tau || !coef4 || T < 1) return fail(HD_E_INVALID, "hd_set_schedule: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    hipFree(h->d_tau); hipFree(h->d_coef);
    h->d_tau = h->d_coef = nullptr;
    h->tau_h.assign(tau, tau + T + 1);
    h->coef_h.assign(coef4, coef4 + (size_t)4 * T);
    HD_TRY(dev_upload(&h->d_tau, h->tau_h));
    HD_TRY(dev_upload(&h->d_coef, h->coef_h));
    h->T = T;
    return HD_OK;
}

extern "C" int hd_sample_loop(hd_handle* h, hd_topology* topo, float* z, const float* context, int mol_shape,
                              int s_hi, int s_lo, const float* raw_x, const float* raw_h, int noise_rows,
                              uint64_t seed, uint64_t sample_id_base, int use_graph, void* stream) {
    HD_TRY(check_ready(h, topo, "hd_sample_loop"));
    if (h->T < 1) return fail(HD_E_STATE, "hd_sample_loop: schedule not set (hd_set_schedule)");
    if (!z) return fail(HD_E_INVALID, "hd_sample_loop: null z");
    if (s_hi > h->T || s_lo < 0 || s_lo > s_hi) return fail(HD_E_INVALID, "hd_sample_loop: need 0 <= s_lo <= s_hi <= T");
    if ((raw_x == nullptr) != (raw_h == nullptr)) return fail(HD_E_INVALID, "hd_sample_loop: raw_x and raw_h go together");
    if (noise_rows != 1 && noise_rows != topo->B) return fail(HD_E_INVALID, "hd_sample_loop: noise_rows must be 1 or B");
    if (h->cfg.context_node_nf > 0 && !context) return fail(HD_E_INVALID, "hd_sample_loop: context required");
    if (!h->cfg.condition_time) return fail(HD_E_INVALID, "hd_sample_loop: needs a time-conditioned model");
    const int mol = (mol_shape < 0 || mol_shape > topo->N) ? topo->N : mol_shape;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int nsteps = s_hi - s_lo;
    if (nsteps == 0) return HD_OK;
    const int T = h->T;
    const uint32_t draw0 = (uint32_t)(T - (s_hi - 1));       // draw index of the first step (draw 0 = z_T)
    const int share = (noise_rows == 1) ? 1 : 0;
    if (!use_graph) {
        for (int k = 0; k < nsteps; ++k) {
            const int sidx = s_hi - 1 - k;
            HD_TRY(forward_impl(h, topo, z, h->d_tau + sidx + 1, 1, context, mol_shape < 0 ? -1 : mol, topo->eps, s));
            NoiseSrc ns = make_noise(raw_x ? raw_x + (size_t)k * noise_rows * mol * 3 : nullptr,
                                     raw_h ? raw_h + (size_t)k * noise_rows * mol * h->F : nullptr, noise_rows, seed,
                                     sample_id_base, draw0 + (uint32_t)k, share);
            HD_TRY(step_impl(h, topo, z, topo->eps, h->d_coef + (size_t)sidx * 4, 1, ns, mol, z, topo->N, nullptr, nullptr, 0, s));
        }
        return HD_OK;
    }
    // hipGraph: capture one step whose step index / draw / time live in device memory, replay it.
    const int was_prof = h->prof;
    h->prof = 0;
    const int s0 = s_hi - 1;
    const uint32_t d0 = draw0;
    HIP_TRY(hipMemcpyAsync(h->d_step, &s0, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_draw, &d0, sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_tcur, h->d_tau + s0 + 1, sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // The legacy NULL stream cannot be captured: everything queued on the caller's stream is complete
    // here (sync above), so run capture + replays on an internal stream and sync it before returning.
    if (s == nullptr) {
        if (!h->own_stream) HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        s = h->own_stream;
    }
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = forward_impl(h, topo, z, h->d_tcur, 1, context, mol_shape < 0 ? -1 : mol, topo->eps, s);
    if (rc == HD_OK) {
        NoiseSrc ns = make_noise(raw_x, raw_h, noise_rows, seed, sample_id_base, draw0, share);
        rc = step_impl(h, topo, z, topo->eps, h->d_coef, 1, ns, mol, z, topo->N, h->d_step, h->d_draw, draw0, s);
    }
    if (rc == HD_OK) hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, s, h->d_step, h->d_draw, h->d_tcur, h->d_tau);
    hipError_t ce = hipStreamEndCapture(s, &graph);
    h->prof = was_prof;
    if (rc != HD_OK) { if (graph) hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) return fail(HD_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ie != hipSuccess) { hipGraphDestroy(graph); return fail(HD_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
    for (int k = 0; k < nsteps; ++k) {
        hipError_t le = hipGraphLaunch(exec, s);
        if (le != hipSuccess) {
            hipGraphExecDestroy(exec); hipGraphDestroy(graph);
            return fail(HD_E_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(le));
        }
    }
    HIP_TRY(hipStreamSynchronize(s));
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    return HD_OK;
}

// ----------------------------------------------------------------------------- host RNG twin

static inline void philox_round_h(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

extern "C" float hd_philox_normal_host(uint64_t seed, uint64_t sample_id, uint32_t draw, uint32_t index) {
    uint32_t c[4] = {index >> 1, draw, (uint32_t)sample_id, (uint32_t)(sample_id >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) { philox_round_h(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = std::sqrt(-2.0f * std::log(u1));
    const float ang = 6.283185307179586f * u2;
    return (index & 1) ? rad * std::sin(ang) : rad * std::cos(ang);
}
