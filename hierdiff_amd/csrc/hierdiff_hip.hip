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
#include <mutex>
#include <vector>

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

// fp32 mode: batches with fewer active rows than this run the node side as three launches (widths >= 128: k_node_split_f32,
// narrower: k_gemm_r16) instead of the fused k_node_f32 (round 5, profiles/r05_f32_fuse_threshold.log, ms per forward chain vs
// fused: B = 128 3.10 vs 3.27, B = 160 3.73 vs 3.78, B = 192 4.28 vs 4.25, B = 224 4.88 vs 4.74)
#define HD_FUSE_MIN_ROWS 5400
// fp16x3 mode: batches with fewer active rows than this run the node update as the three launches of k_node_split.hpp (32 x 32
// output tiles, four K quarters per workgroup) instead of the fused k_node<..., F16> (one workgroup per 32 rows); bit-identical
#define HD_NODE_SPLIT_MAX_ROWS 2048
// topologies with at most this many edge tiles (a third more in fp32) run k_edge_split (one tile per workgroup, columns
// over its four wavefronts) instead of k_edge (one tile per wavefront); bit-identical, see k_edge_split.hpp
#define HD_SPLIT_MAX_TILES 512
// topologies above one whole-tile workgroup per CU and below this many tiles may run k_edge_mixed: a multiple of the CU count
// of whole-tile workgroups plus column-split single-tile workgroups that back-fill (k_edge_split.hpp; rule in launch_edge_h)
#define HD_MIX_MAX_TILES 16384

struct LayerW {                 // float offsets into hd_handle::dw
    size_t ab_img, ab_bias, wrd, w2_img, b2, wa, w3_img, b3, w4_img, b4;
    size_t ab_gimg, w3_gimg, w4_gimg;      // fp32 mode: the same weights as k_gemm_r16 images (node chain below HD_FUSE_MIN_ROWS)
    float ba;
    float w2s, wrmax, wdmax;               // fp16x3: power-of-two scale of the W2 image (else 1); max |w_r|, max |w_d| as packed
    float w3s, w4s, abs_;                  // FP16 node kernel: power-of-two scales of the W3 / W4 / AB images (else 1)
    float w3l1, w4l1, b3max, b4max;        // ... and the constants of its a-priori row bounds (k_node.hpp)
};

struct ProfRec { int fam; hipEvent_t a, b; };

struct hd_handle {
    hd_config cfg;
    int device;
    int H, fin, F, D, NS;       // NS: 32-column sub-tiles per k_gemm workgroup tile
    // arithmetic of the two kernel families, derived from cfg.precision and the width (hd_create):
    //   edge_mode 0 fp32 | 3 fp16x3      node_mode 0 fp32 (k_node_f32 / k_node_split_f32 / k_gemm_r16) | 3 fp16 two-piece
    //   precision 0: 0 / 0;  3: 3 / 3 (H >= 128), else 3 / 0        (1 = bf16x3 and 2 = bf16x6 were retired in ABI 12)
    // `scaled`: the edge model runs in the domain scaled by -log2(e) (fp16x3, silu_scaled in common.hpp)
    int edge_mode, node_mode;
    bool scaled;
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
    // graph-replay state (device words the captured step reads and k_advance moves on)
    int* d_step;
    uint32_t* d_draw;
    float* d_tcur;
    unsigned long long* d_base; // global id of the batch's first sample (Philox stream selector)
    hipStream_t own_stream;     // capture / replay stream used when the caller passes the legacy NULL stream
    hipEvent_t ev_in, ev_out;   // order own_stream against the caller's stream without host syncs
    hipEvent_t ev_last;         // recorded behind the handle's latest graph replay (any stream): the step / draw / time words
    bool ev_last_set;           // above are shared by every topology of the handle, so replays are serialised on it
    unsigned long long weights_gen, sched_gen;   // bumped when the packed weights / schedule tables are re-allocated
    int split_max_tiles;        // HD_SPLIT_MAX_TILES (a measurement build may override it from the environment)
    int fuse_min_rows;          // HD_FUSE_MIN_ROWS
    int node_split_max_rows;    // HD_NODE_SPLIT_MAX_ROWS
    int f32_split;              // 1: the fp32 small-row node chain of widths >= 128 is k_node_split_f32 (0, measurement build: k_gemm_r16)
    int mix_max_tiles;          // HD_MIX_MAX_TILES
    int mix_rounds;             // measurement build: force the number of whole-tile rounds of k_edge_mixed (-1 = rule)
    int n_cu;                   // compute units of the device
#ifdef HD_DEBUG_KERNELS
    long long* d_trace;         // HD_ABLATE bit 16: cycle stamps of the last traced edge launch
    int trace_wg;
    int ablate;                 // HD_ABLATE value read at hd_create
#endif
    // profiling
    int prof;                   // bitmask of kernel families bracketed with events
    int prof_stride;            // bracket every prof_stride-th forward
    long long prof_fwd;
    bool prof_now;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used;
};

// Everything a captured diffusion step has baked in: a cached graph is replayed only when all of it is unchanged.
struct GraphKey {
    const float *raw_x, *raw_h;
    int has_ctx, mol_shape, noise_rows, T, s_hi;       // s_hi: injected-noise offsets are relative to the first step
    uint64_t seed;
    unsigned long long weights_gen, sched_gen;
    bool operator==(const GraphKey& o) const {
        return raw_x == o.raw_x && raw_h == o.raw_h && has_ctx == o.has_ctx && mol_shape == o.mol_shape &&
               noise_rows == o.noise_rows && T == o.T && s_hi == o.s_hi && seed == o.seed && weights_gen == o.weights_gen &&
               sched_gen == o.sched_gen;
    }
};

struct hd_topology {
    hd_handle* h;
    int device;
    char* arena;                               // the one device allocation everything below points into
    int B, N, M, M_pad, E, E_pad, n_tiles, n_wg, n_parts;
    // device tables
    int *node_of, *slot_of, *ei, *ej, *seg_part, *tile_nseg, *pstart, *nvalid;
    int *rptr, *rrows, *sptr, *srows;          // edge rows by receiving / sending node (training kernels)
    std::vector<int>* node_of_host;            // kept for hd_topology_nodes
    float *w2img, *w2timg;                     // training: device-packed images of the current layer's W2 and W2^T
    uint8_t *eseg, *nm_bytes;
    float* nmask;
    // workspace
    float *hbuf, *AB, *AB2, *Tb, *agg, *x0, *xcur, *part, *xpart, *eps;
    float *abmax, *abmax2;                     // fp16x3: [M_pad][2] row maxima of the AB / AB2 buffers
    float *rowinfo;                            // fp16x3, split node chain: [M_pad][2] {max |h_r|, max |[h | agg]_r|} (k_node_split.hpp)
    // hd_sample_loop with use_graph: the captured step works on library-owned copies of z / context so that the
    // instantiated graph survives across calls (the caller's tensors move); one graph per topology
    float *zbuf, *ctxbuf;
    hipGraphExec_t gexec;
    GraphKey gkey;
    // lifetime: the tables arrive in stream order of `stream0` (hd_topology_create_s); `ready` marks their arrival for
    // any other stream a caller launches on.  A topology used on one stream only hands its arena back to the pool
    // (arena_release) with an event instead of a device-wide synchronisation.
    size_t arena_bytes;
    char* staging;                             // pinned host twin of the table region (source of the async upload)
    size_t staging_bytes;
    hipStream_t stream0, last_stream;
    hipEvent_t ready;
    bool multi_stream;
};

// ----------------------------------------------------------------------------- small helpers

// divisor of the neighbour sums: normalization_factor ('sum'), or the number of edge-list entries per receiving node ('mean':
// the reference's list holds all N x N pairs of a molecule, so every node counts the padded N; egnn_new.py:283-288)
static inline float agg_norm(const hd_config& c, const hd_topology* t) {
    return c.aggregation_mean ? (float)t->N : c.normalization_factor;
}

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

static int prepare_kernels(hd_handle* h);
template <int H> static int prepare_edge_bwd_h();

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
    if (cfg->precision != 0 && cfg->precision != 3)
        return fail(HD_E_INVALID, "hd_create: precision must be 0 (fp32) or 3 (fp16x3); 1 (bf16x3) and 2 (bf16x6) were retired in ABI 12");
    if (!cfg->aggregation_mean && !(cfg->normalization_factor != 0.0f)) return fail(HD_E_INVALID, "hd_create: normalization_factor == 0");
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
    {
        const bool wide = cfg->hidden_nf >= 128;
        switch (cfg->precision) {
            case 3: h->edge_mode = 3; h->node_mode = wide ? 3 : 0; break;
            default: h->edge_mode = 0; h->node_mode = 0; break;
        }
        h->scaled = h->edge_mode == 3;
    }
    h->NS = (cfg->hidden_nf == 32) ? 1 : 2;
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
    h->ev_in = h->ev_out = nullptr;
    h->ev_last = nullptr; h->ev_last_set = false;
    h->weights_gen = h->sched_gen = 0;
    h->d_nanflag = nullptr; h->d_nan_events = nullptr; h->d_step = nullptr; h->d_draw = nullptr; h->d_tcur = nullptr;
    h->d_base = nullptr;
    h->split_max_tiles = HD_SPLIT_MAX_TILES;
    h->mix_max_tiles = HD_MIX_MAX_TILES;
    h->fuse_min_rows = HD_FUSE_MIN_ROWS;
    h->node_split_max_rows = HD_NODE_SPLIT_MAX_ROWS;
    h->f32_split = 1;
    h->mix_rounds = -1;
    {
        hipDeviceProp_t prop;
        h->n_cu = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
#ifdef HD_DEBUG_KERNELS
    h->d_trace = nullptr; h->trace_wg = 0;
    { const char* e = getenv("HD_ABLATE"); h->ablate = e ? atoi(e) : 0; }
    { const char* e = getenv("HD_SPLIT_MAX_TILES"); if (e) h->split_max_tiles = atoi(e); }
    { const char* e = getenv("HD_MIX_MAX_TILES"); if (e) h->mix_max_tiles = atoi(e); }
    { const char* e = getenv("HD_FUSE_MIN_ROWS"); if (e) h->fuse_min_rows = atoi(e); }
    { const char* e = getenv("HD_NODE_SPLIT_MAX_ROWS"); if (e) h->node_split_max_rows = atoi(e); }
    { const char* e = getenv("HD_F32_SPLIT"); if (e) h->f32_split = atoi(e); }
    { const char* e = getenv("HD_MIX_ROUNDS"); if (e) h->mix_rounds = atoi(e); }
#endif
    auto create_rest = [&]() -> int {        // every failure below leaves through hd_destroy (frees what exists)
        HD_TRY(dev_alloc(&h->d_nanflag, 1));
        HD_TRY(dev_alloc(&h->d_nan_events, 1));
        HD_TRY(dev_alloc(&h->d_step, 1));
        HD_TRY(dev_alloc(&h->d_draw, 1));
        HD_TRY(dev_alloc(&h->d_tcur, 1));
        HD_TRY(dev_alloc(&h->d_base, 1));
        HIP_TRY(hipMemset(h->d_nanflag, 0, sizeof(int)));
        HIP_TRY(hipMemset(h->d_nan_events, 0, sizeof(long long)));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming));
        return prepare_kernels(h);
    };
    const int r = create_rest();
    if (r != HD_OK) { const std::string keep = g_err; hd_destroy(h); g_err = keep; return r; }
    *out = h;
    return HD_OK;
}

extern "C" int hd_destroy(hd_handle* h) {
    if (!h) return HD_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    hipFree(h->dw); hipFree(h->d_nanflag); hipFree(h->d_nan_events);
    hipFree(h->d_tau); hipFree(h->d_coef); hipFree(h->d_step); hipFree(h->d_draw); hipFree(h->d_tcur); hipFree(h->d_base);
#ifdef HD_DEBUG_KERNELS
    hipFree(h->d_trace);
#endif
    for (auto e : h->pool) hipEventDestroy(e);
    if (h->ev_in) hipEventDestroy(h->ev_in);
    if (h->ev_out) hipEventDestroy(h->ev_out);
    if (h->ev_last) hipEventDestroy(h->ev_last);
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

// B-operand image of k_gemm_r16: [16-column tile][32-wide K chunk][64 lanes][8 p]; lane = (m, slot gs = (odd, hf)),
// value p = W(16 tile + m, 32 chunk + 16 hf + 2 p + odd): the k this lane feeds to instruction p of the chunk.
template <typename Fn>
static void pack_gemm_b16(std::vector<float>& dst, size_t off, int K, int Nc, Fn W) {
    const int nchunk = K / 32;
    for (int ct = 0; ct < Nc / 16; ++ct)
        for (int c = 0; c < nchunk; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int p = 0; p < 8; ++p) {
                    const int m = lane & 15, gs = lane >> 4, odd = gs >> 1, hf = gs & 1;
                    dst[off + (((size_t)ct * nchunk + c) * 64 + lane) * 8 + p] = W(16 * ct + m, 32 * c + 16 * hf + 2 * p + odd);
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

// FP16 node kernel (k_node<..., F16>): [k-step s][column tile ct][hi|lo][64 lanes][8] halves, k = 16s + 8*(lane>>5) + i,
// col = 32ct + (lane&31) - the same byte size as an fp32 image - holding the two FP16 pieces of W x 2^k (largest |element| in
// [2^14, 2^15)).  Returns 2^k;
// *l1 = max over output columns of sum_k |W[col][k]| (the constant of the kernel's row bounds).
template <typename Fn>
static float pack_node_b_f16(std::vector<float>& dstf, size_t off, int K, int Nc, Fn W, float* l1) {
    float wmax = 0.0f, l1max = 0.0f;
    for (int col = 0; col < Nc; ++col) {
        double sum = 0.0;
        for (int k = 0; k < K; ++k) { const float v = std::fabs(W(col, k)); sum += v; if (v > wmax && std::isfinite(v)) wmax = v; }
        l1max = std::max(l1max, (float)(sum * 1.000001));
    }
    if (l1) *l1 = l1max;
    int ex = 0;
    if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
    const float sw = std::ldexp(1.0f, wmax > 0.0f ? std::max(-100, std::min(100, 15 - ex)) : 0);
    _Float16* dst = reinterpret_cast<_Float16*>(dstf.data() + off);
    const int nct = Nc / 32;
    for (int st = 0; st < K / 16; ++st)
        for (int ct = 0; ct < nct; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const float v = W(32 * ct + (lane & 31), 16 * st + 8 * (lane >> 5) + i) * sw;
                    const size_t base = ((size_t)(st * nct + ct) * 2) * 512;
                    const _Float16 hi = (_Float16)v;
                    dst[base + (size_t)lane * 8 + i] = hi;
                    dst[base + 512 + (size_t)lane * 8 + i] = (_Float16)(v - (float)hi);
                }
    return sw;
}

// fused fp32 node kernel (k_node_f32): [32-wide K chunk s][column tile ct][4 q][64 lanes][4 j] floats,
// k = 32s + 16*(lane>>5) + 4q + j, col = 32ct + (lane&31).
template <typename Fn>
static void pack_node_b_f32(std::vector<float>& dst, size_t off, int K, int Nc, Fn W) {
    const int nct = Nc / 32;
    for (int st = 0; st < K / 32; ++st)
        for (int ct = 0; ct < nct; ++ct)
            for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j)
                        dst[off + (((size_t)(st * nct + ct) * 4 + q) * 64 + lane) * 4 + j] =
                            W(32 * ct + (lane & 31), 32 * st + 16 * (lane >> 5) + 4 * q + j);
}

// fp16x3 edge kernel: per K chunk [hi|lo][2 k-steps][H/32 ct][64 lanes][8] halves, k = 32c + 16*(lane>>5) + 8s + i, holding the two
// fp16 pieces of W2 x 2^k, 2^k the power of two that puts the largest
// |element| into [2^14, 2^15) (so heads stay finite and the tails of all but negligible elements normal).  Returns 2^k.
static float pack_edge_w2_f16(std::vector<float>& dstf, size_t off, int H, const float* W2) {
    float wmax = 0.0f;
    for (size_t i = 0; i < (size_t)H * H; ++i) { const float v = std::fabs(W2[i]); if (v > wmax && std::isfinite(v)) wmax = v; }
    int ex = 0;
    if (wmax > 0.0f) (void)std::frexp(wmax, &ex);           // wmax = m 2^ex, m in [0.5, 1)
    const int k = std::max(-100, std::min(100, 15 - ex));   // wmax 2^k in [2^14, 2^15)
    const float sw = std::ldexp(1.0f, wmax > 0.0f ? k : 0);
    _Float16* dst = reinterpret_cast<_Float16*>(dstf.data() + off);
    const int NCT = H / 32;
    for (int c = 0; c < H / 32; ++c)
        for (int st = 0; st < 2; ++st)
            for (int ct = 0; ct < NCT; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int kk = 32 * c + 16 * (lane >> 5) + 8 * st + i;
                        const int col = 32 * ct + (lane & 31);
                        const float v = W2[(size_t)col * H + kk] * sw;
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        const size_t blk = (size_t)c * 32 * H * 2;
                        dst[blk + (((size_t)(0 * 2 + st) * NCT + ct) * 64 + lane) * 8 + i] = hi;
                        dst[blk + (((size_t)(1 * 2 + st) * NCT + ct) * 64 + lane) * 8 + i] = lo;
                    }
    return sw;
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
    const int H = h->H, fin = h->fin;
    const int L = c.n_layers, S = c.inv_sublayers;
    const bool bf = h->scaled;                                // two-way edge modes: scaled domain
    const size_t w2_floats = (size_t)H * H;
    const size_t gx = 2;                                      // node weight images: two fp16 pieces (or one fp32 word) per weight
    const bool nodef32 = h->node_mode == 0;                   // k_node_f32 / k_gemm_r16
    // layout of the packed buffer
    size_t off = 0;
    auto take = [&](size_t cnt) { size_t o = off; off += (cnt + 3) & ~size_t(3); return o; };
    h->embT = take((size_t)fin * H); h->emb_b = take(H); h->outW = take((size_t)fin * H); h->out_b = take(fin);
    h->gcl.assign((size_t)L * S, LayerW());
    h->coord.assign((size_t)L, LayerW());
    for (int i = 0; i < L; ++i) {
        for (int j = 0; j < S; ++j) {
            LayerW& w = h->gcl[(size_t)i * S + j];
            w.ab_img = take((size_t)H * 2 * H * gx / 2); w.ab_bias = take(2 * H); w.wrd = take(2 * H);
            w.w2_img = take(w2_floats); w.b2 = take(H); w.wa = take(H);
            w.w3_img = take((size_t)2 * H * H * gx / 2); w.b3 = take(H); w.w4_img = take((size_t)H * H * gx / 2); w.b4 = take(H);
            w.ab_gimg = w.w3_gimg = w.w4_gimg = 0;
            if (nodef32) { w.ab_gimg = take((size_t)H * 2 * H); w.w3_gimg = take((size_t)2 * H * H); w.w4_gimg = take((size_t)H * H); }
        }
        LayerW& w = h->coord[i];
        w.ab_img = take((size_t)H * 2 * H * gx / 2); w.ab_bias = take(2 * H); w.wrd = take(2 * H);
        w.w2_img = take(w2_floats); w.b2 = take(H); w.wa = take(H);
        w.w3_img = w.b3 = w.w4_img = w.b4 = 0;
        w.ab_gimg = w.w3_gimg = w.w4_gimg = 0;
        if (nodef32) w.ab_gimg = take((size_t)H * 2 * H);
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
    // fp16x3 mode runs the edge model in a scaled domain (see silu_scaled in common.hpp): everything feeding a
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
        w.abs_ = 1.0f;
        if (nodef32) { pack_node_b_f32(pk, w.ab_img, H, 2 * H, wab); pack_gemm_b16(pk, w.ab_gimg, H, 2 * H, wab); }
        else w.abs_ = pack_node_b_f16(pk, w.ab_img, H, 2 * H, wab, nullptr);
        for (int k = 0; k < H; ++k) {
            pk[w.ab_bias + k] = sc(b1[k]);
            pk[w.ab_bias + H + k] = 0.0f;
            pk[w.wrd + k] = sc(W1[(size_t)k * ld + 2 * H]);
            pk[w.wrd + H + k] = sc(W1[(size_t)k * ld + 2 * H + 1]);
        }
        w.wrmax = w.wdmax = 0.0f;
        for (int k = 0; k < H; ++k) {
            w.wrmax = std::max(w.wrmax, std::fabs(pk[w.wrd + k]));
            w.wdmax = std::max(w.wdmax, std::fabs(pk[w.wrd + H + k]));
        }
    };
    // second edge Linear in the image of the handle's edge mode; returns the scale S its accumulators carry (1 but for fp16x3)
    auto pack_w2 = [&](LayerW& w, const float* W2) -> float {
        if (h->edge_mode == 3) return pack_edge_w2_f16(pk, w.w2_img, H, W2);
        pack_edge_w2(pk, w.w2_img, H, W2);
        return 1.0f;
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
            w.w2s = pack_w2(w, W2);
            if (nodef32) {
                pack_node_b_f32(pk, w.w3_img, 2 * H, H, w3);
                pack_node_b_f32(pk, w.w4_img, H, H, w4);
                pack_gemm_b16(pk, w.w3_gimg, 2 * H, H, w3);
                pack_gemm_b16(pk, w.w4_gimg, H, H, w4);
            } else {
                w.w3s = pack_node_b_f16(pk, w.w3_img, 2 * H, H, w3, &w.w3l1);
                w.w4s = pack_node_b_f16(pk, w.w4_img, H, H, w4, &w.w4l1);
                w.b3max = w.b4max = 0.0f;
                for (int k = 0; k < H; ++k) { w.b3max = std::max(w.b3max, std::fabs(b3[k])); w.b4max = std::max(w.b4max, std::fabs(b4[k])); }
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
        w.w2s = pack_w2(w, W6);
        for (int k = 0; k < H; ++k) { pk[w.b2 + k] = sc(b6[k]); pk[w.wa + k] = sc_inv(w7[k]); }
        w.ba = 0.0f;
    }
    if (p - src != n) return fail(HD_E_INVALID, "hd_set_weights: internal layout mismatch");
    if (h->dw_floats != pk.size()) {
        hipFree(h->dw);
        h->dw = nullptr;
        HD_TRY(dev_alloc(&h->dw, pk.size()));
        h->dw_floats = pk.size();
        h->weights_gen++;                      // captured graphs hold the old address
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(h->dw, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    h->weights_set = true;
    return HD_OK;
}

// ----------------------------------------------------------------------------- topology

// Arena pool.  A training loop meets new masks every step, so topologies come and go at step rate: their device
// arena (tables + workspace, ~60 MB at B = 256) and its pinned staging twin are recycled through a small grow-only free
// list instead of hipMalloc / hipFree (which synchronise the device).  A slot carries the event recorded behind the last
// work of its previous owner; the next owner waits for it on the host before overwriting the staging buffer (in a
// steady loop that work finished steps ago).
struct ArenaSlot {
    int device;
    char* dev; size_t dev_bytes;
    char* pinned; size_t pinned_bytes;
    hipEvent_t done;                           // nullptr: nothing pending
};
static std::mutex g_pool_mu;
static std::vector<ArenaSlot> g_pool;
static constexpr size_t kPoolSlots = 12;

static void slot_free(ArenaSlot& sl) {
    if (sl.done) { (void)hipEventSynchronize(sl.done); (void)hipEventDestroy(sl.done); }
    if (sl.dev) (void)hipFree(sl.dev);
    if (sl.pinned) (void)hipHostFree(sl.pinned);
    sl = ArenaSlot{};
}

// smallest pooled slot of this device that holds both sizes, or a new allocation (10 % head room: the next batch's masks
// differ a little); HD_OK with sl filled, the slot's pending event already waited for
static int arena_acquire(int device, size_t dev_bytes, size_t pinned_bytes, ArenaSlot& sl) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); ++i)
            if (g_pool[i].device == device && g_pool[i].dev_bytes >= dev_bytes && g_pool[i].pinned_bytes >= pinned_bytes &&
                (best < 0 || g_pool[i].dev_bytes < g_pool[best].dev_bytes)) best = i;
        if (best >= 0) {
            sl = g_pool[best];
            g_pool.erase(g_pool.begin() + best);
        } else {
            sl = ArenaSlot{};
        }
    }
    if (sl.dev) {
        if (sl.done) { HIP_TRY(hipEventSynchronize(sl.done)); (void)hipEventDestroy(sl.done); sl.done = nullptr; }
        return HD_OK;
    }
    sl.device = device;
    sl.dev_bytes = (dev_bytes + dev_bytes / 10 + 4095) & ~size_t(4095);
    sl.pinned_bytes = (pinned_bytes + pinned_bytes / 10 + 4095) & ~size_t(4095);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&sl.dev), sl.dev_bytes);
    if (e != hipSuccess) {                     // make room: drop the pool and try once more
        (void)hipGetLastError();
        std::vector<ArenaSlot> drop;
        { std::lock_guard<std::mutex> lk(g_pool_mu); drop.swap(g_pool); }
        for (auto& d : drop) slot_free(d);
        e = hipMalloc(reinterpret_cast<void**>(&sl.dev), sl.dev_bytes);
    }
    if (e != hipSuccess) { sl.dev = nullptr; return fail(HD_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipHostMalloc(reinterpret_cast<void**>(&sl.pinned), sl.pinned_bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipFree(sl.dev); sl = ArenaSlot{}; return fail(HD_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    return HD_OK;
}

// The pool is bounded by slots AND by bytes (kPoolBytes of device memory, which torch's caching allocator cannot see or
// reclaim - ADVICE round 4): the oldest slots go first; a slot larger than the whole budget is never kept.
static const size_t kPoolBytes = size_t(1) << 30;
static void arena_release(ArenaSlot sl) {
    std::vector<ArenaSlot> evict;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool.push_back(sl);
        size_t total = 0;
        for (const auto& p : g_pool) total += p.dev_bytes;
        while (!g_pool.empty() && (g_pool.size() > kPoolSlots || total > kPoolBytes)) {
            total -= g_pool.front().dev_bytes;
            evict.push_back(g_pool.front());
            g_pool.erase(g_pool.begin());
        }
    }
    for (auto& e : evict) slot_free(e);
}

extern "C" int hd_arena_pool_trim(void) {
    std::vector<ArenaSlot> drop;
    { std::lock_guard<std::mutex> lk(g_pool_mu); drop.swap(g_pool); }
    for (auto& d : drop) slot_free(d);
    return HD_OK;
}

extern "C" int hd_topology_destroy(hd_topology* t) {
    if (!t) return HD_OK;
    (void)hipSetDevice(t->device);
    ArenaSlot sl{t->device, t->arena, t->arena_bytes, t->staging, t->staging_bytes, nullptr};
    // a captured graph, or launches on several streams: wait for the device (the rare case - a sampling topology lives as
    // long as its model); otherwise an event behind the topology's last work guards the arena's next owner
    bool pooled = t->arena && !t->gexec && !t->multi_stream;
    if (pooled && hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) == hipSuccess) {
        if (hipEventRecord(sl.done, t->last_stream) != hipSuccess) { (void)hipEventDestroy(sl.done); sl.done = nullptr; pooled = false; }
    } else {
        pooled = false;
    }
    if (!pooled) (void)hipDeviceSynchronize();
    if (t->gexec) hipGraphExecDestroy(t->gexec);
    if (t->ready) (void)hipEventDestroy(t->ready);
    if (t->arena) arena_release(sl);                    // tables and workspace live in one allocation
    delete t->node_of_host;
    delete t;
    return HD_OK;
}

// every entry point that launches on a topology passes its stream through here: a stream other than the one the tables
// were uploaded on first waits for their arrival
static inline void topo_use(hd_topology* t, hipStream_t s) {
    if (s != t->last_stream) {
        if (t->ready) (void)hipStreamWaitEvent(s, t->ready, 0);
        t->multi_stream = true;
        t->last_stream = s;
    }
}

// Edge tiles.  The unmasked edges of molecule b, sorted by receiving node, are cut into pieces at MOLECULE-relative
// multiples of 32: full pieces own a 32-row tile, the remainder (< 32 rows) shares a "tail tile" with the remainders
// of neighbouring molecules at 4-row-aligned offsets.  Which edges of a node are summed together (a "part" = the
// node's rows inside one piece) therefore depends on the molecule alone, never on its position in the batch, and the
// edge kernel's per-node sums are invariant under 4-row shifts of a piece (k_edge.hpp) - so a sample comes out
// bit-identical whatever batch / rank / world size it is computed in.  Part ids are node-major
// (a node's parts are contiguous, in piece order = the order the consumers add them); every (tile, segment) carries
// its part id explicitly.
struct TileLayout {
    int M, M_pad, n_tiles, n_wg, n_parts;
    long long E;
    std::vector<int> slot_of, node_of, nvalid, ei, ej, seg_part, tile_nseg, pstart;
    std::vector<int> rptr, rrows, sptr, srows;      // valid edge rows by receiving / sending node (CSR, ascending rows)
    std::vector<uint8_t> eseg;
    std::vector<float> nmask;
};

static int build_layout(const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N, TileLayout& L) {
    if (B < 1 || N < 1) return fail(HD_E_INVALID, "hd_topology_create: B and N must be >= 1");
    if ((long long)B * N > (1LL << 30)) return fail(HD_E_INVALID, "hd_topology_create: B*N too large");
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
    std::vector<int>&slot_of = L.slot_of, &node_of = L.node_of, &nvalid = L.nvalid;
    std::vector<float>& nmask = L.nmask;
    slot_of.assign(BN, -1); node_of.clear(); nvalid.assign(B, 0); nmask.clear();
    for (size_t f = 0; f < BN; ++f) {
        if (active[f]) { slot_of[f] = (int)node_of.size(); node_of.push_back((int)f); nmask.push_back(node_mask[f] ? 1.0f : 0.0f); }
        if (node_mask[f]) nvalid[f / N]++;
    }
    const int M = (int)node_of.size();
    const int M_pad = std::max(128, (M + 127) / 128 * 128);
    nmask.resize(M_pad, 0.0f);
    // per-molecule edge lists (compact node ids), sorted by receiving node then sender
    std::vector<int> mei, mej;
    std::vector<long long> mstart(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < N; ++i) {
            const size_t fi = (size_t)b * N + i;
            if (!active[fi]) continue;
            for (int j = 0; j < N; ++j) {
                const size_t fj = (size_t)b * N + j;
                const bool on = edge_mask ? edge_mask[fi * N + j] != 0 : (node_mask[fi] && node_mask[fj] && i != j);
                if (on) { mei.push_back(slot_of[fi]); mej.push_back(slot_of[fj]); }
            }
        }
        mstart[b + 1] = (long long)mei.size();
    }
    const long long E = (long long)mei.size();
    if (E > (1LL << 30)) return fail(HD_E_INVALID, "hd_topology_create: too many edges");
    // parts per node: the pieces (molecule-relative 32-row blocks) its edge run touches
    std::vector<int>& pstart = L.pstart;
    std::vector<int> first_piece(M, 0);
    pstart.assign(M + 1, 0);
    for (int b = 0; b < B; ++b) {
        const long long e0 = mstart[b], e1 = mstart[b + 1];
        for (long long e = e0; e < e1;) {
            const int node = mei[e];
            long long f = e;
            while (f < e1 && mei[f] == node) ++f;
            first_piece[node] = (int)((e - e0) / 32);
            pstart[node + 1] = (int)((f - 1 - e0) / 32) - first_piece[node] + 1;
            e = f;
        }
    }
    for (int i = 0; i < M; ++i) pstart[i + 1] += pstart[i];
    // tiles
    std::vector<int>&ei = L.ei, &ej = L.ej, &seg_part = L.seg_part, &tile_nseg = L.tile_nseg;
    std::vector<uint8_t>& eseg = L.eseg;
    ei.clear(); ej.clear(); seg_part.clear(); tile_nseg.clear(); eseg.clear();
    auto new_tile = [&]() {
        const int t = (int)tile_nseg.size();
        ei.resize(ei.size() + 32, 0); ej.resize(ej.size() + 32, 0); eseg.resize(eseg.size() + 32, 255);
        seg_part.resize(seg_part.size() + 32, 0); tile_nseg.push_back(0);
        return t;
    };
    auto place = [&](int t, int off, int b, int piece, int cnt) {      // rows [32 piece, 32 piece + cnt) of molecule b
        const long long e0 = mstart[b] + 32LL * piece;
        int prev = -1;
        for (int r = 0; r < cnt; ++r) {
            const int node = mei[e0 + r];
            if (node != prev) {
                seg_part[(size_t)t * 32 + tile_nseg[t]] = pstart[node] + (piece - first_piece[node]);
                tile_nseg[t]++;
                prev = node;
            }
            ei[(size_t)t * 32 + off + r] = node; ej[(size_t)t * 32 + off + r] = mej[e0 + r];
            eseg[(size_t)t * 32 + off + r] = (uint8_t)(tile_nseg[t] - 1);
        }
    };
    int tail_tile = -1, tail_fill = 0;
    for (int b = 0; b < B; ++b) {
        const int Eb = (int)(mstart[b + 1] - mstart[b]);
        const int nfull = Eb / 32, tail = Eb % 32;
        for (int k = 0; k < nfull; ++k) place(new_tile(), 0, b, k, 32);
        if (tail) {
            if (tail_tile < 0 || tail_fill + tail > 32) { tail_tile = new_tile(); tail_fill = 0; }
            place(tail_tile, tail_fill, b, nfull, tail);
            tail_fill = (tail_fill + tail + 3) & ~3;
        }
    }
    L.n_tiles = (int)tile_nseg.size();
    L.n_wg = (L.n_tiles + 3) / 4;
    while ((int)tile_nseg.size() < std::max(1, L.n_wg * 4)) new_tile();     // padding tiles of the last workgroup
    L.M = M; L.M_pad = M_pad; L.E = E; L.n_parts = pstart[M];
    // CSR views of the tiled rows (training kernels): ascending row order makes the per-node sums deterministic
    L.rptr.assign(M + 1, 0); L.sptr.assign(M + 1, 0);
    const size_t rows = ei.size();
    for (size_t r = 0; r < rows; ++r)
        if (eseg[r] != 255) { L.rptr[ei[r] + 1]++; L.sptr[ej[r] + 1]++; }
    for (int i = 0; i < M; ++i) { L.rptr[i + 1] += L.rptr[i]; L.sptr[i + 1] += L.sptr[i]; }
    L.rrows.assign((size_t)E, 0); L.srows.assign((size_t)E, 0);
    {
        std::vector<int> rp(L.rptr.begin(), L.rptr.end() - 1), sp(L.sptr.begin(), L.sptr.end() - 1);
        for (size_t r = 0; r < rows; ++r)
            if (eseg[r] != 255) { L.rrows[rp[ei[r]]++] = (int)r; L.srows[sp[ej[r]]++] = (int)r; }
    }
    return HD_OK;
}

// Host-only view of the tables hd_topology_create builds (no device needed; used by the CPU test tier to check the
// batch-independence of the layout).  counts5 = {active nodes, valid edges, tiles incl. padding tiles, parts, rows};
// any of the output arrays may be NULL (query the sizes first): ei/ej/eseg [rows], seg_part [rows],
// tile_nseg [tiles], pstart [nodes + 1].
extern "C" int hd_topology_layout(const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N, long long* counts5,
                                  int* ei, int* ej, uint8_t* eseg, int* seg_part, int* tile_nseg, int* pstart) {
    if (!node_mask || !counts5) return fail(HD_E_INVALID, "hd_topology_layout: null argument");
    TileLayout L;
    HD_TRY(build_layout(node_mask, edge_mask, B, N, L));
    counts5[0] = L.M; counts5[1] = L.E; counts5[2] = (long long)L.tile_nseg.size(); counts5[3] = L.n_parts;
    counts5[4] = (long long)L.ei.size();
    if (ei) std::copy(L.ei.begin(), L.ei.end(), ei);
    if (ej) std::copy(L.ej.begin(), L.ej.end(), ej);
    if (eseg) std::copy(L.eseg.begin(), L.eseg.end(), eseg);
    if (seg_part) std::copy(L.seg_part.begin(), L.seg_part.end(), seg_part);
    if (tile_nseg) std::copy(L.tile_nseg.begin(), L.tile_nseg.end(), tile_nseg);
    if (pstart) std::copy(L.pstart.begin(), L.pstart.end(), pstart);
    return HD_OK;
}

extern "C" int hd_topology_create_s(hd_handle* h, const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N,
                                    void* stream, hd_topology** out) {
    if (!h || !node_mask || !out) return fail(HD_E_INVALID, "hd_topology_create: null argument");
    hipStream_t s0 = (hipStream_t)stream;
    *out = nullptr;
    TileLayout L;
    HD_TRY(build_layout(node_mask, edge_mask, B, N, L));
    HIP_TRY(hipSetDevice(h->device));
    const size_t BN = (size_t)B * N;
    const int M = L.M, M_pad = L.M_pad, n_tiles = L.n_tiles, n_wg = L.n_wg, n_parts = L.n_parts;
    const long long E = L.E;
    const int E_pad = (int)L.ei.size();
    std::vector<int>&slot_of = L.slot_of, &node_of = L.node_of, &nvalid = L.nvalid, &ei = L.ei, &ej = L.ej;
    std::vector<int>&seg_part = L.seg_part, &tile_nseg = L.tile_nseg, &pstart = L.pstart;
    std::vector<uint8_t>& eseg = L.eseg;
    std::vector<float>& nmask = L.nmask;
    std::vector<uint8_t> nm_bytes(BN);
    for (size_t f = 0; f < BN; ++f) nm_bytes[f] = node_mask[f] ? 1 : 0;

    hd_topology* t = new hd_topology();
    std::memset(t, 0, sizeof(*t));
    t->h = h; t->device = h->device; t->B = B; t->N = N; t->M = M; t->M_pad = M_pad; t->E = (int)E; t->E_pad = E_pad;
    t->n_tiles = n_tiles; t->n_wg = n_wg; t->n_parts = n_parts;
    const int H = h->H;
    // ONE device allocation per topology: [index tables | zero-filled activation workspace], filled by ONE host-to-device
    // copy and ONE memset.  (Round 2 made ~35 hipMalloc / hipMemset / hipMemcpy calls per topology: 1.2 - 3.5 ms for a new
    // batch of masks, which a training step pays every time - ADVICE round 2.)
    std::vector<char> blob;
    auto stage = [&](const void* src, size_t bytes) {               // 256-byte aligned slot in the table region
        const size_t off = (blob.size() + 255) & ~size_t(255);
        blob.resize(off + std::max<size_t>(bytes, 4), 0);
        if (bytes) std::memcpy(blob.data() + off, src, bytes);
        return off;
    };
    auto stage_v = [&](const auto& v) { return stage(v.data(), v.size() * sizeof(v[0])); };
    const size_t o_node_of = stage_v(node_of), o_slot_of = stage_v(slot_of), o_ei = stage_v(ei), o_ej = stage_v(ej);
    const size_t o_seg_part = stage_v(seg_part), o_tile_nseg = stage_v(tile_nseg), o_pstart = stage_v(pstart), o_nvalid = stage_v(nvalid);
    const size_t o_eseg = stage_v(eseg), o_nm = stage_v(nm_bytes), o_nmask = stage_v(nmask);
    const size_t o_rptr = stage_v(L.rptr), o_rrows = stage_v(L.rrows), o_sptr = stage_v(L.sptr), o_srows = stage_v(L.srows);
    const size_t table_bytes = (blob.size() + 255) & ~size_t(255);
    size_t ws_floats = 0;
    auto carve = [&](size_t count) { const size_t o = ws_floats; ws_floats += (std::max<size_t>(count, 1) + 63) & ~size_t(63); return o; };
    const size_t f_h = carve((size_t)M_pad * H), f_AB = carve((size_t)M_pad * 2 * H), f_AB2 = carve((size_t)M_pad * 2 * H);
    const size_t f_Tb = carve((size_t)M_pad * H), f_agg = carve((size_t)M_pad * H);       // fp32 node chain of small batches; training
    const size_t f_x0 = carve((size_t)M_pad * 4), f_xcur = carve((size_t)M_pad * 4);
    const size_t f_part = carve((size_t)std::max(1, n_parts) * H), f_xpart = carve((size_t)std::max(1, n_parts) * 4);
    const size_t f_abmax = carve((size_t)M_pad * 2), f_abmax2 = carve((size_t)M_pad * 2), f_rowinfo = carve((size_t)M_pad * 2);
    const size_t f_eps = carve(BN * h->D), f_z = carve(BN * h->D), f_ctx = carve(BN * (size_t)std::max(1, h->cfg.context_node_nf));
    const size_t f_w2 = carve((size_t)H * H * 3 / 2), f_w2t = carve((size_t)H * H * 3 / 2);      // fp32 / fp16 images (H^2 floats) + room behind them: the fp16x3 training forward parks its image scalars there
    auto build = [&]() -> int {
        ArenaSlot sl;
        HD_TRY(arena_acquire(h->device, table_bytes + ws_floats * sizeof(float), table_bytes, sl));
        t->arena = sl.dev; t->arena_bytes = sl.dev_bytes; t->staging = sl.pinned; t->staging_bytes = sl.pinned_bytes;
        t->stream0 = t->last_stream = s0;
        char* base = t->arena;
        // no host wait from here on: tables pinned staging -> device and the workspace fill in stream order of s0
        std::memcpy(t->staging, blob.data(), blob.size());
        HIP_TRY(hipMemcpyAsync(base, t->staging, blob.size(), hipMemcpyHostToDevice, s0));
        // zero-filled workspace: pad rows stay zero for the lifetime of the topology (kernels never write them)
        HIP_TRY(hipMemsetAsync(base + table_bytes, 0, ws_floats * sizeof(float), s0));
        HIP_TRY(hipEventCreateWithFlags(&t->ready, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(t->ready, s0));
        auto I = [&](size_t off) { return reinterpret_cast<int*>(base + off); };
        t->node_of = I(o_node_of); t->slot_of = I(o_slot_of); t->ei = I(o_ei); t->ej = I(o_ej); t->seg_part = I(o_seg_part);
        t->tile_nseg = I(o_tile_nseg); t->pstart = I(o_pstart); t->nvalid = I(o_nvalid);
        t->eseg = reinterpret_cast<uint8_t*>(base + o_eseg); t->nm_bytes = reinterpret_cast<uint8_t*>(base + o_nm);
        t->nmask = reinterpret_cast<float*>(base + o_nmask);
        t->rptr = I(o_rptr); t->rrows = I(o_rrows); t->sptr = I(o_sptr); t->srows = I(o_srows);
        float* ws = reinterpret_cast<float*>(base + table_bytes);
        t->hbuf = ws + f_h; t->AB = ws + f_AB; t->AB2 = ws + f_AB2; t->Tb = ws + f_Tb; t->agg = ws + f_agg; t->x0 = ws + f_x0;
        t->xcur = ws + f_xcur; t->part = ws + f_part; t->xpart = ws + f_xpart; t->eps = ws + f_eps; t->zbuf = ws + f_z;
        t->abmax = ws + f_abmax; t->abmax2 = ws + f_abmax2; t->rowinfo = ws + f_rowinfo;
        t->ctxbuf = ws + f_ctx; t->w2img = ws + f_w2; t->w2timg = ws + f_w2t;
        t->node_of_host = new std::vector<int>(node_of);
        return HD_OK;
    };
    const int r = build();
    if (r != HD_OK) { const std::string keep = g_err; hd_topology_destroy(t); g_err = keep; return r; }
    *out = t;
    return HD_OK;
}

// the tables are on the device when this returns (any stream may launch on the topology)
extern "C" int hd_topology_create(hd_handle* h, const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N,
                                  hd_topology** out) {
    HD_TRY(hd_topology_create_s(h, node_mask, edge_mask, B, N, nullptr, out));
    HIP_TRY(hipStreamSynchronize(nullptr));
    hipEvent_t ev = (*out)->ready;              // arrived: no stream has anything to wait for
    (*out)->ready = nullptr;
    if (ev) (void)hipEventDestroy(ev);
    return HD_OK;
}

__global__ void k_widen_index(const int* src, long long* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// hd_topology_nodes for a device consumer: the compact node order as int64 flat indices b*N + n, written in stream order
extern "C" int hd_topology_nodes_device(hd_topology* t, long long* index, void* stream) {
    if (!t || !index) return fail(HD_E_INVALID, "hd_topology_nodes_device: null argument");
    HIP_TRY(hipSetDevice(t->device));
    topo_use(t, (hipStream_t)stream);
    if (t->M > 0) hipLaunchKernelGGL(k_widen_index, dim3((t->M + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->node_of, index, t->M);
    HIP_TRY(hipGetLastError());
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
    if (cat && epi == EPI_RANK1_SILU) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_RANK1_SILU, true>), grid, block, 0, s, g);
    else if (epi == EPI_BIAS_MASK) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_BIAS_MASK, false>), grid, block, 0, s, g);
    else if (cat && epi == EPI_BIAS) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_BIAS, true>), grid, block, 0, s, g);
    else if (cat) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_BIAS_SILU, true>), grid, block, 0, s, g);
    else if (epi == EPI_BIAS_SILU) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_BIAS_SILU, false>), grid, block, 0, s, g);
    else if (epi == EPI_BIAS) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_BIAS, false>), grid, block, 0, s, g);
    else if (epi == EPI_EGCL_PRE) hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_EGCL_PRE, false>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((k_gemm<WM, WN, CN, EPI_RESID_MASK, false>), grid, block, 0, s, g);
}

template <int RT>
static void launch_r16(int epi, bool agg, const R16Args& g, hipStream_t s) {
    const int wpt = g.Nc >= 128 ? 8 : (g.Nc >> 4);
    const dim3 grid(((g.M + 16 * RT - 1) / (16 * RT)) * (g.Nc / (16 * wpt)) * g.n_img), block(512);
    const int lds = 16 * RT * (g.K + 4) * 4;
    if (agg) hipLaunchKernelGGL((k_gemm_r16<EPI_BIAS_SILU, true, RT>), grid, block, lds, s, g);
    else if (epi == EPI_BIAS) hipLaunchKernelGGL((k_gemm_r16<EPI_BIAS, false, RT>), grid, block, lds, s, g);
    else if (epi == EPI_BIAS_SILU) hipLaunchKernelGGL((k_gemm_r16<EPI_BIAS_SILU, false, RT>), grid, block, lds, s, g);
    else hipLaunchKernelGGL((k_gemm_r16<EPI_RESID_MASK, false, RT>), grid, block, lds, s, g);
}

static void gemm_r16(hd_handle* h, int epi, bool agg, const R16Args& g, hipStream_t s) {
    ProfScope ps(h, s, 1);
    launch_r16<1>(epi, agg, g, s);      // 32-row workgroups (RT = 2) were measured: slower up to B = 64, equal above (profiles/r03_r16_sweep2.log)
}

// Fused node update in two-piece FP16 arithmetic (k_node<..., F16>), 32-row workgroups.
template <int H>
static int node_lds_bytes(bool upd, int np = 2) {        // region 0: the two fp16 pieces of X; region 1: those of T / the fp32 staging tile
    const int r1 = std::max(32 * (H + 8) * 2 * np, 32 * (H + 4) * 4);
    return 32 * ((upd ? 2 * H : H) + 8) * 2 * np + r1;
}

template <int H>
static int node_f32_lds_bytes(bool upd) { return 32 * ((upd ? 2 * H : H) + 4) * 4 + 32 * (H + 4) * 4; }

template <int H, int NW>
static void launch_node_hw(bool upd, int nab, int mode, const NodeArgs& a, hipStream_t s) {      // mode: 0 fp32, 3 fp16 two-piece
    const int nrt = (a.M + 31) / 32;
    const dim3 grid(8 * ((nrt + 7) / 8)), block(64 * NW);
    if (mode == 0) {
        const int ldsf = node_f32_lds_bytes<H>(upd);
        if (!upd) hipLaunchKernelGGL((k_node_f32<H, NW, false, 1>), grid, block, ldsf, s, a);
        else if (nab == 1) hipLaunchKernelGGL((k_node_f32<H, NW, true, 1>), grid, block, ldsf, s, a);
        else hipLaunchKernelGGL((k_node_f32<H, NW, true, 2>), grid, block, ldsf, s, a);
        return;
    }
    if constexpr (H >= 128) {                              // two-piece FP16 (fp16x3 mode; narrower widths run the fp32 node kernels)
        const int lds = node_lds_bytes<H>(upd);
        if (!upd) hipLaunchKernelGGL((k_node<H, NW, false, 1, 2, true>), grid, block, lds, s, a);
        else if (nab == 1) hipLaunchKernelGGL((k_node<H, NW, true, 1, 2, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_node<H, NW, true, 2, 2, true>), grid, block, lds, s, a);
    }
}

template <int H, int NW>
static int prepare_node_hw() {
    HIP_TRY(hipFuncSetAttribute((const void*)k_node_f32<H, NW, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, node_f32_lds_bytes<H>(false)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_node_f32<H, NW, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, node_f32_lds_bytes<H>(true)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_node_f32<H, NW, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, node_f32_lds_bytes<H>(true)));
    if constexpr (H >= 128) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_node<H, NW, false, 1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, node_lds_bytes<H>(false)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_node<H, NW, true, 1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, node_lds_bytes<H>(true)));
        HIP_TRY(hipFuncSetAttribute((const void*)k_node<H, NW, true, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, node_lds_bytes<H>(true)));
    }
    return HD_OK;
}

// wavefronts per 32-row workgroup: one 32-column tile of an H-wide output each, at most 8 (two per SIMD)
template <int H>
static void launch_node_h(bool upd, int nab, int mode, const NodeArgs& a, hipStream_t s) {
    launch_node_hw<H, (H / 32 < 8 ? H / 32 : 8)>(upd, nab, mode, a, s);
}

template <int H>
static int prepare_node_h() { return prepare_node_hw<H, (H / 32 < 8 ? H / 32 : 8)>(); }

static void node_update(hd_handle* h, bool upd, int nab, const NodeArgs& a, hipStream_t s) {
    ProfScope ps(h, s, 1);
    const int mode = h->node_mode;
    switch (h->H) {
        case 32: launch_node_h<32>(upd, nab, mode, a, s); break;
        case 64: launch_node_h<64>(upd, nab, mode, a, s); break;
        case 128: launch_node_h<128>(upd, nab, mode, a, s); break;
        default: launch_node_h<256>(upd, nab, mode, a, s); break;
    }
}

// fp16x3, few rows: the node update as three launches of 32 x 32 output tiles (k_node_split.hpp), bit-identical to the fused kernel
template <int H, int PH, int CTW>
static void launch_node_split_hc(const NodeSplitArgs& a, hipStream_t s) {
    const int nrt = (a.M + 31) / 32;
    const int per = (PH == 3 ? 2 * H / 32 * a.n_img : H / 32) / CTW;
    const int lds = node_split_lds_bytes<H, PH, CTW>();
    hipLaunchKernelGGL((k_node_split<H, PH, CTW>), dim3(8 * ((nrt + 7) / 8) * per), dim3(256), lds, s, a);
}
// one 32-column tile per workgroup: two (half the workgroups, half the redundant operand loads; same bits) were measured slower
// at every size from 24 to 64 molecules (profiles/r05_node_split_sweep.log: 0.85 / 0.86 / 0.94 / 0.96 of the fused kernel's forward
// time with one tile, 0.90 / 0.91 / 0.95 / 0.97 with two)
template <int H, int PH>
static void launch_node_split_h(const NodeSplitArgs& a, hipStream_t s) { launch_node_split_hc<H, PH, 1>(a, s); }
template <int PH>
static void launch_node_split(hd_handle* h, const NodeSplitArgs& a, hipStream_t s) {
    ProfScope ps(h, s, 1);
    if (h->H == 128) launch_node_split_h<128, PH>(a, s); else launch_node_split_h<256, PH>(a, s);
}
template <int H, int PH, int CTW>
static int prepare_node_split_one() {
    const int lds = node_split_lds_bytes<H, PH, CTW>();
    HIP_TRY(hipFuncSetAttribute((const void*)k_node_split<H, PH, CTW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return HD_OK;
}
// exact fp32, widths >= 128, below HD_FUSE_MIN_ROWS rows: the same three launches on k_node_f32's arithmetic (k_node_split_f32)
template <int H, int PH>
static void launch_node_split_f32_h(const NodeSplitArgs& a, hipStream_t s) {
    const int nrt = (a.M + 31) / 32;
    const int per = (PH == 3 ? 2 * H / 32 * a.n_img : H / 32);
    const int lds = node_split_f32_lds_bytes<H, PH>();
    hipLaunchKernelGGL((k_node_split_f32<H, PH>), dim3(8 * ((nrt + 7) / 8) * per), dim3(256), lds, s, a);
}
template <int PH>
static void launch_node_split_f32(hd_handle* h, const NodeSplitArgs& a, hipStream_t s) {
    ProfScope ps(h, s, 1);
    if (h->H == 128) launch_node_split_f32_h<128, PH>(a, s); else launch_node_split_f32_h<256, PH>(a, s);
}
template <int H, int PH>
static int prepare_node_split_f32_one() {
    const int lds = node_split_f32_lds_bytes<H, PH>();
    HIP_TRY(hipFuncSetAttribute((const void*)k_node_split_f32<H, PH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return HD_OK;
}

template <int H>
static int prepare_node_split_h() {
    HD_TRY((prepare_node_split_f32_one<H, 1>())); HD_TRY((prepare_node_split_f32_one<H, 2>())); HD_TRY((prepare_node_split_f32_one<H, 3>()));
    HD_TRY((prepare_node_split_one<H, 1, 1>())); HD_TRY((prepare_node_split_one<H, 2, 1>())); HD_TRY((prepare_node_split_one<H, 3, 1>()));
    return HD_OK;
}

template <int H>
static int edge_lds_bytes() {       // dynamic part: W2 double buffer + wave scratch (w_r/w_d/b2/wa are static)
    return (2 * 32 * H + 2 * H + 4 * 136) * 4;
}

#ifdef HD_DEBUG_KERNELS
// Measurement build only (python -m hierdiff_amd.build --debug-kernels): HD_ABLATE=<bits> selects an ablated
// instantiation of the H=256 GCL edge kernel; bit 16 records per-wave cycle stamps (hd_debug_edge_trace).
// Not compiled into the product library: the variants change results and allocate on first use.
extern "C" int hd_debug_edge_trace(hd_handle* h, long long* out, int max_wg) {
    if (!h || !h->d_trace) return 0;
    const int n = std::min(max_wg, h->trace_wg);
    hipDeviceSynchronize();
    hipMemcpy(out, h->d_trace, sizeof(long long) * 32 * n, hipMemcpyDeviceToHost);
    return n;
}

// stamps of the last k_node_f32 launch: [workgroup < 512][8 waves][HD_NTRACE_STAMPS] (scratch/node_trace.py)
extern "C" int hd_debug_node_trace(long long* out, int max_ll) {
    hipDeviceSynchronize();
    const int n = std::min(max_ll, 512 * 8 * HD_NTRACE_STAMPS);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(hd_ntrace), sizeof(long long) * n) != hipSuccess) return 0;
    return n;
}

template <int PREC>
static bool launch_edge_ablated(hd_handle* h, const EdgeArgs& a, hipStream_t s) {
    const int lds = edge_lds_bytes<256>();
    const dim3 grid(a.n_wg), block(256);
    auto run = [&](auto Abl) {
        constexpr int ABL = decltype(Abl)::value;
        hipFuncSetAttribute((const void*)k_edge<256, false, PREC, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        EdgeArgs b = a;
        if constexpr (ABL & 16) {
            if (!h->d_trace) hipMalloc(reinterpret_cast<void**>(&h->d_trace), sizeof(long long) * 32 * 4096);
            if (a.n_wg > 4096) return;
            b.trace = h->d_trace; h->trace_wg = a.n_wg;
        }
        hipLaunchKernelGGL((k_edge<256, false, PREC, ABL>), grid, block, lds, s, b);
    };
    switch (h->ablate) {
        case 1: run(std::integral_constant<int, 1>{}); return true;
        case 2: run(std::integral_constant<int, 2>{}); return true;
        case 4: run(std::integral_constant<int, 4>{}); return true;
        case 5: run(std::integral_constant<int, 5>{}); return true;
        case 8: run(std::integral_constant<int, 8>{}); return true;
        case 12: run(std::integral_constant<int, 12>{}); return true;
        case 13: run(std::integral_constant<int, 13>{}); return true;
        case 15: run(std::integral_constant<int, 15>{}); return true;
        case 16: run(std::integral_constant<int, 16>{}); return true;
        case 18: run(std::integral_constant<int, 18>{}); return true;
        case 20: run(std::integral_constant<int, 20>{}); return true;
        case 24: run(std::integral_constant<int, 24>{}); return true;
        case 30: run(std::integral_constant<int, 30>{}); return true;
        case 32: run(std::integral_constant<int, 32>{}); return true;
        case 64: run(std::integral_constant<int, 64>{}); return true;
        case 128: run(std::integral_constant<int, 128>{}); return true;
        case 1024: run(std::integral_constant<int, 1024>{}); return true;
        case 1025: run(std::integral_constant<int, 1025>{}); return true;
        default: return false;
    }
}
#else
extern "C" int hd_debug_edge_trace(hd_handle*, long long*, int) { return 0; }      // product build: nothing is traced
#endif

// Which kernel family runs a list of n_tiles edge tiles at width >= 128 in arithmetic `mode` (launch_edge_h below and
// hd_edge_layer_save_rows ask the same question): the column-split kernel, the mix of whole and column-split tiles, or k_edge.
static bool edge_runs_split(const hd_handle* h, int n_tiles, int mode) {
    return n_tiles > 0 && n_tiles <= (mode == 0 ? h->split_max_tiles + h->split_max_tiles / 3 : h->split_max_tiles);
}
static bool edge_runs_mixed(const hd_handle* h, int n_tiles, int mode) {
    const int per_round = 4 * h->n_cu;
    const int left = n_tiles - (n_tiles / per_round) * per_round;
    const bool pays = left > 0 && left * 10 <= (mode == 0 ? 29 : 10) * h->n_cu;
    return n_tiles > per_round && n_tiles < h->mix_max_tiles && (pays || h->mix_rounds >= 0);
}

// `mode_override` >= 0 selects the arithmetic of THIS launch (0 fp32, 3 fp16x3) instead of the handle's: the opt-in fp16x3
// forward of the training path (hd_edge_layer_forward_s) on a handle whose other kernels stay exact fp32
template <int H>
static int launch_edge_h(hd_handle* h, bool coord, const EdgeArgs& a, hipStream_t s, int mode_override = -1) {
    const int lds = edge_lds_bytes<H>();
    // kernel family of this launch: 0 fp32, 3 fp16x3
    const int prec = mode_override >= 0 ? mode_override : h->edge_mode;
    const dim3 grid(a.n_wg), block(256);
    if constexpr (H >= 128) {
        if (a.dscal) {
            // training forward in fp16x3 arithmetic: the whole-tile kernel on the unscaled parameters (HD_EDGE_UNSCALED), keeping pre2 or not
            if (a.pre2) {
                if (coord) hipLaunchKernelGGL((k_edge<H, true, 3, HD_EDGE_SAVE | HD_EDGE_UNSCALED>), grid, block, lds, s, a);
                else hipLaunchKernelGGL((k_edge<H, false, 3, HD_EDGE_SAVE | HD_EDGE_UNSCALED>), grid, block, lds, s, a);
            } else {
                if (coord) hipLaunchKernelGGL((k_edge<H, true, 3, HD_EDGE_UNSCALED>), grid, block, lds, s, a);
                else hipLaunchKernelGGL((k_edge<H, false, 3, HD_EDGE_UNSCALED>), grid, block, lds, s, a);
            }
            return HD_OK;
        }
    }
    if (a.pre2) {
        // training forward that keeps pre2 for the backward pass (HD_EDGE_SAVE): always the whole-tile kernel, whose accumulator
        // layout is the saved layout (the caller offers the buffer only where this kernel would run anyway: edge_layer_saves)
        if (coord) hipLaunchKernelGGL((k_edge<H, true, 0, HD_EDGE_SAVE>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_edge<H, false, 0, HD_EDGE_SAVE>), grid, block, lds, s, a);
        return HD_OK;
    }
#ifdef HD_DEBUG_KERNELS
    if constexpr (H == 256) {
        if (h->ablate && !coord && (prec == 3 ? launch_edge_ablated<3>(h, a, s) : launch_edge_ablated<0>(h, a, s))) return HD_OK;
    }
#endif
    if constexpr (H >= 128) {
        // at most 512 tiles: one tile per workgroup, columns split over its four wavefronts (k_edge_split.hpp; bit-identical
        // to k_edge in every precision mode, a quarter of the serial MFMA chain per wavefront)
        const int mode = prec;
        // measured break-even (profiles/history/r02_split_sweep.log): between 490 and 654 tiles in the 16-bit split modes, between 654 and 870
        // in fp32 (the longer MFMA chain has more to gain from the split)
        if (edge_runs_split(h, a.n_tiles, mode)) {
            const dim3 sgrid(a.n_tiles);
            if (mode == 0) {
                if (coord) hipLaunchKernelGGL((k_edge_split<H, true, 0>), sgrid, block, 0, s, a);
                else hipLaunchKernelGGL((k_edge_split<H, false, 0>), sgrid, block, 0, s, a);
            } else {
                if (coord) hipLaunchKernelGGL((k_edge_split<H, true, 3>), sgrid, block, 0, s, a);
                else hipLaunchKernelGGL((k_edge_split<H, false, 3>), sgrid, block, 0, s, a);
            }
            return HD_OK;
        }
    }
    if constexpr (H >= 128) {
        // between one whole-tile workgroup per CU and HD_MIX_MAX_TILES: R * n_cu whole-tile workgroups (every CU the same
        // number) + the remaining tiles as column-split workgroups that back-fill (k_edge_mixed; bit-identical per tile)
        const int mode = prec;
        const int per_round = 4 * h->n_cu;
        // Measured (profiles/r03_mix_sweep*.log, ms per forward, plain -> mixed): fp32 B = 40 1.89 -> 1.45, 64 1.93 -> 1.88,
        // 96 2.66 -> 2.60, 128 3.29 -> 3.17, 160 4.15 -> 3.89, 192 4.79 -> 4.40, 256 5.41 -> 5.51; the 16-bit split modes gain only
        // while few tiles are left over (B = 40: -14 %; B = 64 ... 256: +2 ... +8 %).  A column-split tile costs about 1.5 x a
        // whole one in SIMD time, so the mix pays when it replaces a badly filled last round: at most 2.9 left-over tiles per
        // CU in fp32, 1.0 in fp16x3.
        int R = a.n_tiles / per_round;
        if (edge_runs_mixed(h, a.n_tiles, mode)) {
            if (h->mix_rounds >= 0) R = std::min(R, h->mix_rounds);
            EdgeArgs m = a;
            m.n_wg = R * h->n_cu;
            const dim3 mgrid(m.n_wg + (a.n_tiles - 4 * m.n_wg));
            const int ldsm = std::max(edge_lds_bytes<H>(), edge_split_lds_bytes<H, 0>());
            if (mode == 0) {
                if (coord) hipLaunchKernelGGL((k_edge_mixed<H, true, 0>), mgrid, block, ldsm, s, m);
                else hipLaunchKernelGGL((k_edge_mixed<H, false, 0>), mgrid, block, ldsm, s, m);
            } else {
                if (coord) hipLaunchKernelGGL((k_edge_mixed<H, true, 3>), mgrid, block, ldsm, s, m);
                else hipLaunchKernelGGL((k_edge_mixed<H, false, 3>), mgrid, block, ldsm, s, m);
            }
            return HD_OK;
        }
    }
    if (prec == 3) {
        if (coord) hipLaunchKernelGGL((k_edge<H, true, 3>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_edge<H, false, 3>), grid, block, lds, s, a);
    } else {
        if (coord) hipLaunchKernelGGL((k_edge<H, true, 0>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((k_edge<H, false, 0>), grid, block, lds, s, a);
    }
    return HD_OK;
}

// Raise the dynamic-LDS limit of the edge kernels for this device (done once, at hd_create: it is
// not allowed while a stream is capturing).
template <int H>
static int prepare_edge_h() {
    const int lds = edge_lds_bytes<H>();
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, true, 0, HD_EDGE_SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, false, 0, HD_EDGE_SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if constexpr (H >= 128) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, true, 3, HD_EDGE_SAVE | HD_EDGE_UNSCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, false, 3, HD_EDGE_SAVE | HD_EDGE_UNSCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, true, 3, HD_EDGE_UNSCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge<H, false, 3, HD_EDGE_UNSCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const int m0 = std::max(edge_lds_bytes<H>(), edge_split_lds_bytes<H, 0>());
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_mixed<H, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, m0));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_mixed<H, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, m0));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_mixed<H, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, m0));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_mixed<H, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, m0));
    }
    return HD_OK;
}

static int prepare_kernels(hd_handle* h) {
    switch (h->H) {
        case 32: HD_TRY(prepare_node_h<32>()); HD_TRY(prepare_edge_bwd_h<32>()); return prepare_edge_h<32>();
        case 64: HD_TRY(prepare_node_h<64>()); HD_TRY(prepare_edge_bwd_h<64>()); return prepare_edge_h<64>();
        case 128: HD_TRY(prepare_node_h<128>()); HD_TRY(prepare_node_split_h<128>()); HD_TRY(prepare_edge_bwd_h<128>()); return prepare_edge_h<128>();
        default: HD_TRY(prepare_node_h<256>()); HD_TRY(prepare_node_split_h<256>()); HD_TRY(prepare_edge_bwd_h<256>()); return prepare_edge_h<256>();
    }
}

static int edge(hd_handle* h, bool coord, const EdgeArgs& a, hipStream_t s, int mode_override = -1) {
    if (a.n_wg == 0) return HD_OK;
    ProfScope ps(h, s, 0);
    switch (h->H) {
        case 32: return launch_edge_h<32>(h, coord, a, s, mode_override);
        case 64: return launch_edge_h<64>(h, coord, a, s, mode_override);
        case 128: return launch_edge_h<128>(h, coord, a, s, mode_override);
        default: return launch_edge_h<256>(h, coord, a, s, mode_override);
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
            // fp16x3, few rows: the AB-only launch of k_node_split accumulates the row maxima of AB with an atomic max
            a.zero_max = (h->node_mode == 3 && M < h->node_split_max_rows && h->edge_mode == 3) ? t->abmax : nullptr;
            a.M = M; a.N = t->N; a.D = h->D; a.F = h->F; a.C = c.context_node_nf; a.H = H;
            a.t_stride = (t_numel == 1) ? 0 : 1; a.cond_time = c.condition_time;
            const long long total = (long long)M * (H / 4);
            hipLaunchKernelGGL(k_node_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        }
        const float range = c.coords_range / (float)c.n_layers;
        const int S = c.inv_sublayers;
        // AB of the layer about to run is produced by the previous node update (or, for the very first layer, by an
        // AB-only launch); `ab_cur` is the buffer the next edge kernel reads
        const float* ab_cur = t->AB;
        auto node_args = [&]() {
            NodeArgs a;
            std::memset(&a, 0, sizeof(a));
            a.h_in = t->hbuf; a.h_out = t->hbuf; a.part = t->part; a.pstart = t->pstart; a.nmask = t->nmask;
            a.norm = agg_norm(c, t); a.M = M;
            return a;
        };
        auto set_ab = [&](NodeArgs& a, int q, const LayerW& nw, float* dst) {
            a.ABimg[q] = W + nw.ab_img; a.ABbias[q] = W + nw.ab_bias; a.ABout[q] = dst;
            a.abinv[q] = h->node_mode == 3 ? 1.0f / nw.abs_ : 1.0f;
            // fp16x3 at widths >= 128: the fused node kernel leaves the row maxima of what it writes (narrower widths: k_ab_rowmax)
            a.ABmax[q] = (h->edge_mode == 3 && h->node_mode != 0) ? (dst == t->AB ? t->abmax : t->abmax2) : nullptr;
        };
        // fp32 mode, few rows: the fused kernel's serial chain per 32-row workgroup (~50 us) is not hidden by other workgroups
        // (60 of them at B = 64), the k_agg + 3 x k_gemm chain spreads the same work over 64 x 64 tiles.  Both paths are
        // bit-identical (same MFMA order per output element, bias added after the contraction), so a sample's bits do
        // not depend on which one its batch size selects (test_fp32_node_paths_agree_bitwise).
        const bool fused = h->node_mode != 0 || M >= h->fuse_min_rows;
        auto r16_args = [&]() {
            R16Args g;
            std::memset(&g, 0, sizeof(g));
            g.A = t->hbuf; g.lda = H; g.K1 = H; g.K = H; g.M = M; g.n_img = 1; g.nmask = t->nmask; g.norm = agg_norm(c, t);
            return g;
        };
        auto ab_r16 = [&](int nab, const LayerW* const* nxt, float* const* dst) {       // AB_q = h [W1a | W1b]_q^T + [b1 | 0]
            R16Args g = r16_args();
            g.Nc = 2 * H; g.ldc = 2 * H; g.n_img = nab;
            for (int q = 0; q < nab; ++q) { g.Bimg[q] = W + nxt[q]->ab_gimg; g.bias[q] = W + nxt[q]->ab_bias; g.C[q] = dst[q]; }
            gemm_r16(h, EPI_BIAS, false, g, s);
        };
        // fp16x3, few rows: the fused kernel's chain per 32-row workgroup (20 - 27 us, a weight stream through one CU) is not hidden
        // when only a few workgroups exist; k_node_split spreads every phase's columns over workgroups (bit-identical)
        const bool nsplit = h->node_mode == 3 && M < h->node_split_max_rows;
        // exact fp32, widths >= 128, below HD_FUSE_MIN_ROWS: k_node_split_f32 (bit-identical to k_node_f32) instead of k_gemm_r16
        const bool fsplit = h->node_mode == 0 && M < h->fuse_min_rows && H >= 128 && h->f32_split;
        auto split_args = [&]() {
            NodeSplitArgs a;
            std::memset(&a, 0, sizeof(a));
            a.h_in = t->hbuf; a.h_out = t->hbuf; a.part = t->part; a.pstart = t->pstart; a.nmask = t->nmask; a.norm = agg_norm(c, t);
            a.T = t->Tb; a.rowinfo = t->rowinfo; a.M = M; a.n_img = 1;
            return a;
        };
        auto split_ab = [&](NodeSplitArgs& a, int q, const LayerW& nw, float* dst) {
            a.Wimg[q] = W + nw.ab_img; a.bias[q] = W + nw.ab_bias; a.winv[q] = 1.0f / nw.abs_; a.ABout[q] = dst;
            a.ABmax[q] = h->edge_mode == 3 ? (dst == t->AB ? t->abmax : t->abmax2) : nullptr;
        };
        {
            const LayerW* first[2] = {&h->gcl[0], nullptr};
            float* dst[2] = {t->AB, nullptr};
            if (nsplit) {
                NodeSplitArgs a = split_args();
                split_ab(a, 0, h->gcl[0], t->AB);
                a.upd = 0;                   // (the row maxima start from the zeros k_node_init left)
                launch_node_split<3>(h, a, s);
            } else if (fsplit) {
                NodeSplitArgs a = split_args();
                a.Wimg[0] = W + h->gcl[0].ab_img; a.bias[0] = W + h->gcl[0].ab_bias; a.ABout[0] = t->AB; a.upd = 0;
                launch_node_split_f32<3>(h, a, s);
            } else if (fused) {
                NodeArgs a = node_args();
                set_ab(a, 0, h->gcl[0], t->AB);
                node_update(h, false, 1, a, s);
            } else {
                ab_r16(1, first, dst);
            }
        }
        for (int i = 0; i < c.n_layers; ++i) {
            for (int j = 0; j <= c.inv_sublayers; ++j) {
                const bool coord = (j == c.inv_sublayers);
                const LayerW& w = coord ? h->coord[i] : h->gcl[(size_t)i * c.inv_sublayers + j];
                EdgeArgs e;
                std::memset(&e, 0, sizeof(e));
                e.AB = ab_cur; e.wrd = W + w.wrd; e.W2img = W + w.w2_img; e.b2 = W + w.b2; e.wa = W + w.wa;
                e.w2s_inv = 1.0f / w.w2s; e.wrmax = w.wrmax; e.wdmax = w.wdmax;
                const bool fused_max = h->edge_mode == 3 && h->node_mode != 0;
                e.abmax = (fused_max && ab_cur == t->AB2) ? t->abmax2 : t->abmax;
                if (h->edge_mode == 3 && !fused_max) {  // fp16x3: the per-node part of the activation bound (k_edge.hpp)
                    ProfScope ps(h, s, 2);
                    AbMaxArgs am{ab_cur, t->abmax, M, H};
                    hipLaunchKernelGGL(k_ab_rowmax, dim3((M + 3) / 4), dim3(256), 0, s, am);
                }
                e.ei = t->ei; e.ej = t->ej; e.eseg = t->eseg; e.seg_part = t->seg_part; e.tile_nseg = t->tile_nseg;
                e.xcur = t->xcur; e.x0 = t->x0; e.part = coord ? t->xpart : t->part; e.ba = w.ba;
                e.norm_constant = c.norm_constant; e.coords_range = range; e.attention = c.attention;
                e.use_tanh = c.tanh; e.n_tiles = t->n_tiles; e.n_wg = t->n_wg;
                HD_TRY(edge(h, coord, e, s));
                if (!coord) {
                    // node update + the first edge Linear of the layer(s) that follow: AB of the next sub-layer, or (after the
                    // block's last GCL) of the coordinate layer and of the next block's first GCL
                    const LayerW* nxt[2] = {nullptr, nullptr};
                    float* dst[2] = {t->AB, t->AB2};
                    int nab = 1;
                    if (j + 1 < S) {
                        nxt[0] = &h->gcl[(size_t)i * S + j + 1];
                    } else {
                        nxt[0] = &h->coord[i];
                        if (i + 1 < c.n_layers) { nxt[1] = &h->gcl[(size_t)(i + 1) * S]; nab = 2; }
                    }
                    if (nsplit) {
                        NodeSplitArgs a = split_args();
                        a.w3l1 = w.w3l1; a.w4l1 = w.w4l1; a.b3max = w.b3max; a.b4max = w.b4max;
                        NodeSplitArgs p1 = a, p2 = a, p3 = a;
                        p1.Wimg[0] = W + w.w3_img; p1.bias[0] = W + w.b3; p1.winv[0] = 1.0f / w.w3s;
                        launch_node_split<1>(h, p1, s);
                        p2.Wimg[0] = W + w.w4_img; p2.bias[0] = W + w.b4; p2.winv[0] = 1.0f / w.w4s;
                        p3.n_img = nab; p3.upd = 1;
                        for (int q = 0; q < nab; ++q) { split_ab(p3, q, *nxt[q], dst[q]); p2.zero_max[q] = p3.ABmax[q]; }
                        launch_node_split<2>(h, p2, s);
                        launch_node_split<3>(h, p3, s);
                    } else if (fsplit) {
                        NodeSplitArgs p1 = split_args(), p2 = split_args(), p3 = split_args();
                        p1.Wimg[0] = W + w.w3_img; p1.bias[0] = W + w.b3;
                        launch_node_split_f32<1>(h, p1, s);
                        p2.Wimg[0] = W + w.w4_img; p2.bias[0] = W + w.b4;
                        launch_node_split_f32<2>(h, p2, s);
                        p3.n_img = nab; p3.upd = 1;
                        for (int q = 0; q < nab; ++q) { p3.Wimg[q] = W + nxt[q]->ab_img; p3.bias[q] = W + nxt[q]->ab_bias; p3.ABout[q] = dst[q]; }
                        launch_node_split_f32<3>(h, p3, s);
                    } else if (fused) {
                        NodeArgs a = node_args();
                        a.W3img = W + w.w3_img; a.b3 = W + w.b3; a.W4img = W + w.w4_img; a.b4 = W + w.b4;
                        if (h->node_mode == 3) {
                            a.w3inv = 1.0f / w.w3s; a.w4inv = 1.0f / w.w4s; a.w3l1 = w.w3l1; a.w4l1 = w.w4l1; a.b3max = w.b3max; a.b4max = w.b4max;
                        }
                        for (int q = 0; q < nab; ++q) set_ab(a, q, *nxt[q], dst[q]);
                        node_update(h, true, nab, a, s);
                    } else {
                        R16Args g = r16_args();
                        g.part = t->part; g.pstart = t->pstart; g.K1 = H; g.K = 2 * H; g.Nc = H;
                        g.Bimg[0] = W + w.w3_gimg; g.bias[0] = W + w.b3; g.C[0] = t->Tb; g.ldc = H;
                        gemm_r16(h, EPI_BIAS_SILU, true, g, s);                      // T = silu([h | agg] W3^T + b3)
                        g = r16_args();
                        g.A = t->Tb; g.Bimg[0] = W + w.w4_gimg; g.bias[0] = W + w.b4; g.C[0] = t->hbuf; g.ldc = H; g.Nc = H;
                        gemm_r16(h, EPI_RESID_MASK, false, g, s);                    // h = (h + T W4^T + b4) mask
                        ab_r16(nab, nxt, dst);
                    }
                    ab_cur = t->AB;
                } else {
                    ProfScope ps(h, s, 2);
                    XupdArgs x;
                    x.part = t->xpart; x.pstart = t->pstart; x.nmask = t->nmask; x.xcur = t->xcur;
                    x.norm = agg_norm(c, t); x.M = M;
                    hipLaunchKernelGGL(k_xupd, dim3((M + 255) / 256), dim3(256), 0, s, x);
                    ab_cur = t->AB2;                        // next block's first GCL (written with the last node update)
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
    topo_use(topo, (hipStream_t)stream);
    return forward_impl(h, topo, xh, t, t_numel, context, mol_shape, out, (hipStream_t)stream);
}

extern "C" int hd_nan_events(hd_handle* h, void* stream, long long* count) {
    if (!h || !count) return fail(HD_E_INVALID, "hd_nan_events: null argument");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(count, h->d_nan_events, sizeof(long long), hipMemcpyDeviceToHost));
    return HD_OK;
}

// ----------------------------------------------------------------------------- training primitives (fp32)

extern "C" int hd_topology_nodes(const hd_topology* t, int* node_of) {
    if (!t || !node_of) return fail(HD_E_INVALID, "hd_topology_nodes: null argument");
    std::copy(t->node_of_host->begin(), t->node_of_host->end(), node_of);
    return HD_OK;
}

template <int H>
static int edge_bwd_lds_bytes() { return (2 * 32 * H + 4 * 288) * 4; }      // two weight chunks + per-wave scratch

template <int H>
static int prepare_edge_bwd_h() {
    const int lds = edge_bwd_lds_bytes<H>();
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge_bwd<H, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge_bwd<H, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge_bwd<H, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute((const void*)k_edge_bwd<H, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if constexpr (H >= 128) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_bwd<H, false, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_edge_bwd<H, true, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    return HD_OK;
}

template <int H>
static void launch_edge_bwd_h(bool coord, int stage, int prec, const EdgeBwdArgs& a, int n_wg, hipStream_t s) {
    const dim3 grid(n_wg), block(256);
    if (stage == 0 && a.pre2) {                     // pre2 kept by the forward pass: stage A loads it (one kernel for both arithmetics)
        const int ldss = 4 * 288 * 4;
        if (coord) hipLaunchKernelGGL((k_edge_bwd<H, true, 0, 0, true>), grid, block, ldss, s, a);
        else hipLaunchKernelGGL((k_edge_bwd<H, false, 0, 0, true>), grid, block, ldss, s, a);
        return;
    }
    if constexpr (H >= 128) {
        if (prec == 3 && stage == 1) {              // fp16x3 contraction of stage B (16-wide chunks of 16 H floats)
            const int lds = (2 * 16 * H + 4 * 288) * 4;
            if (coord) hipLaunchKernelGGL((k_edge_bwd<H, true, 1, 3>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((k_edge_bwd<H, false, 1, 3>), grid, block, lds, s, a);
            return;
        }
    }
    const int lds = edge_bwd_lds_bytes<H>();
    if (!coord && stage == 0) hipLaunchKernelGGL((k_edge_bwd<H, false, 0>), grid, block, lds, s, a);
    else if (!coord) hipLaunchKernelGGL((k_edge_bwd<H, false, 1>), grid, block, lds, s, a);
    else if (stage == 0) hipLaunchKernelGGL((k_edge_bwd<H, true, 0>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((k_edge_bwd<H, true, 1>), grid, block, lds, s, a);
}

static void launch_edge_bwd(hd_handle* h, bool coord, int stage, int prec, const EdgeBwdArgs& a, int n_wg, hipStream_t s) {
    switch (h->H) {
        case 32: launch_edge_bwd_h<32>(coord, stage, 0, a, n_wg, s); break;
        case 64: launch_edge_bwd_h<64>(coord, stage, 0, a, n_wg, s); break;
        case 128: launch_edge_bwd_h<128>(coord, stage, prec, a, n_wg, s); break;
        default: launch_edge_bwd_h<256>(coord, stage, prec, a, n_wg, s); break;
    }
}

static int check_train(hd_handle* h, hd_topology* t, const char* who) {
    if (!h || !t) return fail(HD_E_INVALID, std::string(who) + ": null handle/topology");
    if (t->h != h) return fail(HD_E_INVALID, std::string(who) + ": topology belongs to another handle");
    if (h->cfg.precision != 0) return fail(HD_E_STATE, std::string(who) + ": training primitives need precision 0 (exact fp32)");
    return HD_OK;
}

// Floats of the workspace the fp16x3 backward wants (hd_edge_layer_backward_s, precision 3): image scalars, per-workgroup maxima of
// |G2| and |P| (read by hd_dw2_f16 at offsets 4 and 4 + n), per-row maxima of G2.  *n_wg = n, the number of per-workgroup maxima.
extern "C" long long hd_edge_layer_f16ws_floats(hd_handle* h, hd_topology* t, int* n_wg) {
    if (!h || !t || t->h != h) return 0;
    if (n_wg) *n_wg = t->n_wg;
    return 4 + 2 * (long long)t->n_wg + (long long)t->n_wg * 4 * 32;
}

// Rows of the pre2 buffer a training forward may keep for its backward pass (hd_edge_layer_forward_s), or 0 where keeping does
// not pay: batches small enough for the column-split edge kernel keep their faster forward and recompute.
extern "C" long long hd_edge_layer_save_rows(hd_handle* h, hd_topology* t, int precision) {
    if (!h || !t || t->h != h || t->n_wg == 0 || t->M == 0) return 0;
    if (precision != 0 && precision != 3) return 0;
    const int mode = h->H >= 128 ? precision : 0;
    // (the mix of whole and column-split tiles gains 2-5 % on a forward; the kept pre-activations save the backward a whole contraction:
    // only the pure column-split regime - B <= 18 at N = 30, where that forward is 2 x faster - keeps recomputing)
    if (h->H >= 128 && edge_runs_split(h, t->n_tiles, mode)) return 0;
    return (long long)t->n_wg * 4 * 32 + 32;        // one tile more than the table has: the fp16x3 forward parks its image scalars there
}

extern "C" int hd_edge_layer_forward_s(hd_handle* h, hd_topology* t, int coord, int precision, const float* AB, const float* x,
                                       const float* x0, const float* wrd, const float* W2, const float* b2,
                                       const float* wa, const float* ba, float* out, float* pre2, void* stream) {
    HD_TRY(check_train(h, t, "hd_edge_layer_forward"));
    if (precision != 0 && precision != 3)
        return fail(HD_E_INVALID, "hd_edge_layer_forward_s: precision must be 0 (fp32) or 3 (fp16x3); 2 (bf16x6) was retired in ABI 12");
    const bool f16 = precision == 3 && h->H >= 128;         // narrower widths run the exact-fp32 kernels, as in sampling
    topo_use(t, (hipStream_t)stream);
    if (!AB || !x || !x0 || !wrd || !W2 || !b2 || !wa || !out) return fail(HD_E_INVALID, "hd_edge_layer_forward: null tensor");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const hd_config& c = h->cfg;
    const int H = h->H, M = t->M;
    const int ow = coord ? 4 : H;
    // (k_agg below writes every row of `out`; only a layer without edges needs the fill)
    if (t->n_wg == 0 || M == 0) {
        HIP_TRY(hipMemsetAsync(out, 0, (size_t)std::max(1, M) * ow * sizeof(float), s));
        return HD_OK;
    }
    EdgeArgs e;
    std::memset(&e, 0, sizeof(e));
    if (f16) {
        // fp16 image of the parameter and its scalars (behind the image: the buffer holds 1.5 H^2 floats), row maxima of the AB rows
        // (with a kept pre2 the scalars live in the buffer's extra tile, where the backward call finds them again)
        float* scal = pre2 ? pre2 + (size_t)t->n_wg * 128 * H : t->w2img + (size_t)H * H;
        hipLaunchKernelGGL(k_f16_prep, dim3(1), dim3(1024), 0, s, W2, wrd, scal, H);
        hipLaunchKernelGGL((k_pack_w2_f16<false>), dim3((H * H / 8 + 255) / 256), dim3(256), 0, s, W2, (const float*)scal,
                           reinterpret_cast<f16x8*>(t->w2img), H);
        AbMaxArgs am{AB, t->abmax, M, H};
        hipLaunchKernelGGL(k_ab_rowmax, dim3((M + 3) / 4), dim3(256), 0, s, am);
        e.dscal = scal; e.abmax = t->abmax;
    } else hipLaunchKernelGGL((k_pack_w2<false>), dim3((H * H + 255) / 256), dim3(256), 0, s, W2, t->w2img, H);
    e.AB = AB; e.wrd = wrd; e.W2img = t->w2img; e.b2 = b2; e.wa = wa;
    e.w2s_inv = 1.0f; e.wrmax = e.wdmax = 0.0f;
    e.ei = t->ei; e.ej = t->ej; e.eseg = t->eseg; e.seg_part = t->seg_part; e.tile_nseg = t->tile_nseg;
    e.xcur = x; e.x0 = x0; e.part = coord ? t->xpart : t->part; e.ba = 0.0f; e.ba_ptr = ba;
    e.norm_constant = c.norm_constant; e.coords_range = c.coords_range / (float)c.n_layers; e.attention = c.attention;
    e.use_tanh = c.tanh; e.n_tiles = t->n_tiles; e.n_wg = t->n_wg;
    e.pre2 = pre2;
    HD_TRY(edge(h, coord != 0, e, s, f16 ? 3 : 0));
    AggArgs ag;
    ag.part = coord ? t->xpart : t->part; ag.pstart = t->pstart; ag.agg = out; ag.norm = agg_norm(c, t);
    ag.M = M; ag.H = ow;
    const long long total = (long long)M * (ow / 4);
    hipLaunchKernelGGL(k_agg, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ag);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_edge_layer_forward(hd_handle* h, hd_topology* t, int coord, const float* AB, const float* x,
                                     const float* x0, const float* wrd, const float* W2, const float* b2,
                                     const float* wa, const float* ba, float* out, void* stream) {
    return hd_edge_layer_forward_s(h, t, coord, 0, AB, x, x0, wrd, W2, b2, wa, ba, out, nullptr, stream);
}

extern "C" int hd_edge_layer_backward_s(hd_handle* h, hd_topology* t, int coord, int precision, const float* AB, const float* x,
                                      const float* x0, const float* wrd, const float* W2, const float* b2,
                                      const float* wa, const float* ba, const float* gout, const float* pre2, float* f16ws,
                                      float* G2, float* P, float* G1,
                                      float* escal, float* colpart, float* bapart, float* b2part, float* wrdpart,
                                      float* dAB, float* dx, float* dx0, void* stream) {
    HD_TRY(check_train(h, t, "hd_edge_layer_backward"));
    if (precision != 0 && precision != 3)
        return fail(HD_E_INVALID, "hd_edge_layer_backward_s: precision must be 0 (fp32) or 3 (fp16x3); 2 (bf16x6) was retired in ABI 12");
    if (precision == 3 && (!pre2 || !f16ws))
        return fail(HD_E_INVALID, "hd_edge_layer_backward_s: precision 3 (fp16x3) needs the kept pre2 and the f16ws workspace");
    const bool f16 = precision == 3 && h->H >= 128;
    topo_use(t, (hipStream_t)stream);
    if (!AB || !x || !x0 || !wrd || !W2 || !b2 || !wa || !gout || !G2 || !P || !G1 || !escal || !colpart || !bapart ||
        !b2part || !wrdpart || !dAB || !dx || !dx0)
        return fail(HD_E_INVALID, "hd_edge_layer_backward: null tensor");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const hd_config& c = h->cfg;
    const int H = h->H, M = t->M;
    const int tiles = std::max(1, t->n_wg * 4);
    if (t->n_wg == 0 || M == 0) {
        HIP_TRY(hipMemsetAsync(dAB, 0, (size_t)std::max(1, M) * 2 * H * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(dx, 0, (size_t)std::max(1, M) * 4 * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(dx0, 0, (size_t)std::max(1, M) * 4 * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(escal, 0, (size_t)tiles * 32 * 8 * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(colpart, 0, (size_t)tiles * H * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(bapart, 0, (size_t)tiles * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(b2part, 0, (size_t)tiles * H * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(wrdpart, 0, (size_t)tiles * 2 * H * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(G2, 0, (size_t)tiles * 32 * H * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(P, 0, (size_t)tiles * 32 * H * sizeof(float), s));
        HIP_TRY(hipMemsetAsync(G1, 0, (size_t)tiles * 32 * H * sizeof(float), s));
        return HD_OK;
    }
    // every output row is written by the kernels below (all tiles of the padded table exist; the CSR sums and k_edge_dx
    // cover every active node), so nothing is cleared first; escal[:, 0:4] is only defined (and only read) in coordinate layers
    // (bapart of a coordinate layer - no attention bias - is zeroed by stage A itself)
    // (stage A streams the W2 image only when it has to recompute pre2)
    const float* f16scal = f16 ? pre2 + (size_t)t->n_wg * 128 * H : nullptr;     // {2^k, 2^-k, ..} of W2, left by the precision-3 forward
    if (f16) {
        hipLaunchKernelGGL((k_pack_w2_f16c<true>), dim3((H * H / 8 + 255) / 256), dim3(256), 0, s, W2, f16scal,
                           reinterpret_cast<f16x8*>(t->w2timg), H);
    } else {
        if (!pre2) hipLaunchKernelGGL(k_pack_w2_both, dim3((H * H + 255) / 256), dim3(256), 0, s, W2, t->w2img, t->w2timg, H);
        else hipLaunchKernelGGL((k_pack_w2<true>), dim3((H * H + 255) / 256), dim3(256), 0, s, W2, t->w2timg, H);
    }
    EdgeBwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.AB = AB; a.wrd = wrd; a.b2 = b2; a.wa = wa; a.ei = t->ei; a.ej = t->ej; a.eseg = t->eseg; a.xcur = x; a.x0 = x0;
    a.ba_ptr = ba; a.norm_constant = c.norm_constant; a.coords_range = c.coords_range / (float)c.n_layers;
    a.inv_norm = 1.0f / agg_norm(c, t); a.attention = c.attention; a.use_tanh = c.tanh; a.n_tiles = t->n_tiles;
    a.gin = gout; a.G2 = G2; a.escal = escal; a.colpart = colpart; a.bapart = bapart; a.Pout = P; a.G1 = G1;
    a.b2part = b2part; a.wrdpart = wrdpart; a.pre2 = pre2;
    if (f16) {      // f16ws: {2^k, 2^-k, -, -} | per-workgroup maxima of |G2|, |P| | row maxima of G2  (hd_edge_layer_f16ws_floats)
        a.w2scal = f16scal; a.g2wgmax = f16ws + 4; a.pwgmax = f16ws + 4 + t->n_wg; a.g2max = f16ws + 4 + 2 * (size_t)t->n_wg;
    }
    const int prec = f16 ? 3 : 0;
    a.Wimg = t->w2img;
    launch_edge_bwd(h, coord != 0, 0, prec, a, t->n_wg, s);
    a.Wimg = t->w2timg;
    launch_edge_bwd(h, coord != 0, 1, prec, a, t->n_wg, s);
    CsrSumArgs cs;
    std::memset(&cs, 0, sizeof(cs));
    cs.G = G1; cs.out = dAB; cs.M = M; cs.H = H; cs.ldo = 2 * H;
    const long long total = (long long)M * (H / 4);
    cs.ptr = t->rptr; cs.rows = t->rrows; cs.col0 = 0;                   // rows by receiving node -> dA
    cs.ptr_b = t->sptr; cs.rows_b = t->srows; cs.col0_b = H;            // rows by sending node   -> dB   (blockIdx.y = 1)
    hipLaunchKernelGGL(k_csr_sum, dim3((unsigned)((total + 255) / 256), 2), dim3(256), 0, s, cs);
    EdgeDxArgs d;
    d.escal = escal; d.ei = t->ei; d.ej = t->ej; d.rptr = t->rptr; d.rrows = t->rrows; d.sptr = t->sptr; d.srows = t->srows;
    d.xcur = x; d.x0 = x0; d.dx = dx; d.dx0 = dx0; d.norm_constant = c.norm_constant; d.M = M; d.coord = coord ? 1 : 0;
    hipLaunchKernelGGL(k_edge_dx, dim3((M + 3) / 4), dim3(256), 0, s, d);          // one wavefront per node
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_edge_layer_backward(hd_handle* h, hd_topology* t, int coord, const float* AB, const float* x,
                                      const float* x0, const float* wrd, const float* W2, const float* b2,
                                      const float* wa, const float* ba, const float* gout, float* G2, float* P, float* G1,
                                      float* escal, float* colpart, float* bapart, float* b2part, float* wrdpart,
                                      float* dAB, float* dx, float* dx0, void* stream) {
    return hd_edge_layer_backward_s(h, t, coord, 0, AB, x, x0, wrd, W2, b2, wa, ba, gout, nullptr, nullptr, G2, P, G1, escal, colpart,
                                    bapart, b2part, wrdpart, dAB, dx, dx0, stream);
}

// ----------------------------------------------------------------------------- stage-2 layer E_GCL (forward)

struct hd_egcl {
    hd_egcl_config cfg;
    int device, H, De, ctx, NS;
    long long n_weights;
    bool weights_set;
    float* dw;
    size_t dw_floats;
    // float offsets into dw
    size_t ab_img, ab_bias, w_r, w_e, w_c, w1e_img, zero_bias, w2_img, b2, wa, ba, wc1_img, bc1, wc2, we1_img, be1, w_er,
        we2_img, be2, wn1_img, bn1, wn2_img, bn2, ones;
    // widths 128 / 256 (round 5): the three node-level contractions also as k_node_f32 fragment images, for k_node_split_f32
    size_t ab_nimg, wn1_nimg, wn2_nimg;
    bool node_split;
};

struct hd_egcl_graph {
    hd_egcl* g;
    int device, M, E, Mp, Ep;
    int *row, *col, *cptr, *crows;
    float *hin, *hres, *x4, *AB, *agg, *xagg, *Tn, *ea, *T1, *P, *M1, *C1, *geo, *trans, *ones;
};

static long long egcl_weight_count(const hd_egcl_config& c) {
    const long long H = c.hidden_nf, De = c.edges_in_d, ctx = c.context_nf;
    long long n = H * (2 * H + 1 + De + ctx) + H + H * H + H;                 // mes_mlp
    if (c.edge_update) n += H * (H + 1 + De) + H + H * H + H;                 // edge_mlp
    n += H * 2 * H + H + H * H + H;                                           // node_mlp
    if (c.coord_update) n += H * H + H + H;                                   // coord_mlp
    if (c.attention) n += H + 1;                                              // att_mlp
    return n;
}

extern "C" long long hd_egcl_weight_count(const hd_egcl* g) { return g ? g->n_weights : 0; }

extern "C" int hd_egcl_create(const hd_egcl_config* cfg, int device, hd_egcl** out) {
    if (!cfg || !out) return fail(HD_E_INVALID, "hd_egcl_create: null argument");
    *out = nullptr;
    const int H = cfg->hidden_nf;
    if (H != 32 && H != 64 && H != 128 && H != 256) return fail(HD_E_INVALID, "hd_egcl_create: hidden_nf must be 32, 64, 128 or 256");
    if (cfg->edges_in_d < 0 || (cfg->edges_in_d != H && cfg->edges_in_d >= 32))
        return fail(HD_E_INVALID, "hd_egcl_create: edges_in_d must be hidden_nf (edge features) or < 32 (scalar edge attributes)");
    if (cfg->edge_update && cfg->edges_in_d != H) return fail(HD_E_INVALID, "hd_egcl_create: edge_update needs edges_in_d == hidden_nf");
    if (cfg->context_nf < 0 || cfg->context_nf > H) return fail(HD_E_INVALID, "hd_egcl_create: bad context_nf");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, "hd_egcl_create: no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    hd_egcl* g = new hd_egcl();
    std::memset(g, 0, sizeof(*g));
    g->cfg = *cfg; g->device = device; g->H = H; g->De = cfg->edges_in_d; g->ctx = cfg->context_nf;
    g->NS = (H == 32) ? 1 : 2;
    g->n_weights = egcl_weight_count(*cfg);
    // k_node_split_f32 (the layer's node side at widths 128 / 256) needs more than 64 KB of dynamic LDS: raise the limit on this device
    if (H == 128) { const int r = prepare_node_split_h<128>(); if (r != HD_OK) { delete g; return r; } }
    if (H == 256) { const int r = prepare_node_split_h<256>(); if (r != HD_OK) { delete g; return r; } }
    *out = g;
    return HD_OK;
}

extern "C" int hd_egcl_destroy(hd_egcl* g) {
    if (!g) return HD_OK;
    (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();
    hipFree(g->dw);
    delete g;
    return HD_OK;
}

extern "C" int hd_egcl_set_weights(hd_egcl* g, const float* blob, long long n, int on_device, void* stream) {
    if (!g || !blob) return fail(HD_E_INVALID, "hd_egcl_set_weights: null argument");
    if (n != g->n_weights)
        return fail(HD_E_INVALID, "hd_egcl_set_weights: expected " + std::to_string(g->n_weights) + " values, got " + std::to_string(n));
    HIP_TRY(hipSetDevice(g->device));
    std::vector<float> host;
    const float* src = blob;
    if (on_device) {
        host.resize((size_t)n);
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        HIP_TRY(hipMemcpy(host.data(), blob, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
        src = host.data();
    }
    const hd_egcl_config& c = g->cfg;
    const int H = g->H, De = g->De, ctx = g->ctx, WN = g->NS;
    const bool wide = De == H;
    size_t off = 0;
    auto take = [&](size_t cnt) { size_t o = off; off += (cnt + 3) & ~size_t(3); return o; };
    g->ab_img = take((size_t)H * 2 * H); g->ab_bias = take(2 * H); g->w_r = take(H); g->w_e = take((size_t)std::max(1, De) * H);
    g->w_c = take((size_t)std::max(1, ctx) * H); g->w1e_img = take((size_t)H * H); g->zero_bias = take(2 * H);
    g->w2_img = take((size_t)H * H); g->b2 = take(H); g->wa = take(H); g->ba = take(4);
    g->wc1_img = take((size_t)H * H); g->bc1 = take(H); g->wc2 = take(H);
    g->we1_img = take((size_t)2 * H * H); g->be1 = take(H); g->w_er = take(H); g->we2_img = take((size_t)H * H); g->be2 = take(H);
    g->wn1_img = take((size_t)2 * H * H); g->bn1 = take(H); g->wn2_img = take((size_t)H * H); g->bn2 = take(H);
    g->node_split = (H == 128 || H == 256);
    if (g->node_split) { g->ab_nimg = take((size_t)H * 2 * H); g->wn1_nimg = take((size_t)2 * H * H); g->wn2_nimg = take((size_t)H * H); }
    std::vector<float> pk(off, 0.0f);
    const float* p = src;
    auto next = [&](size_t cnt) { const float* q = p; p += cnt; return q; };
    {   // mes_mlp.0 [H][2H + 1 + De + ctx]: columns [source(H) | target(H) | radial | edge_attr(De) | context(ctx)] (gcl.py:92-98)
        const int ld = 2 * H + 1 + De + ctx;
        const float* W1 = next((size_t)H * ld); const float* b1 = next(H);
        pack_gemm_b(pk, g->ab_img, H, 2 * H, WN, [&](int col, int k) {
            return (col < H) ? W1[(size_t)col * ld + k] : W1[(size_t)(col - H) * ld + H + k]; });
        if (g->node_split) pack_node_b_f32(pk, g->ab_nimg, H, 2 * H, [&](int col, int k) {
            return (col < H) ? W1[(size_t)col * ld + k] : W1[(size_t)(col - H) * ld + H + k]; });
        for (int k = 0; k < H; ++k) {
            pk[g->ab_bias + k] = b1[k];
            pk[g->w_r + k] = W1[(size_t)k * ld + 2 * H];
            for (int d = 0; d < De && !wide; ++d) pk[g->w_e + (size_t)d * H + k] = W1[(size_t)k * ld + 2 * H + 1 + d];
            for (int d = 0; d < ctx; ++d) pk[g->w_c + (size_t)d * H + k] = W1[(size_t)k * ld + 2 * H + 1 + De + d];
        }
        if (wide) pack_gemm_b(pk, g->w1e_img, H, H, WN, [&](int col, int k) { return W1[(size_t)col * ld + 2 * H + 1 + k]; });
        const float* W2 = next((size_t)H * H); const float* b2 = next(H);
        pack_gemm_b(pk, g->w2_img, H, H, WN, [&](int col, int k) { return W2[(size_t)col * H + k]; });
        std::copy(b2, b2 + H, pk.begin() + g->b2);
    }
    if (c.edge_update) {   // edge_mlp.0 [H][H + 1 + De]: columns [edge_feat(H) | radial | edge_attr(De)] (gcl.py:111)
        const int ld = H + 1 + De;
        const float* We1 = next((size_t)H * ld); const float* be1 = next(H);
        pack_gemm_b(pk, g->we1_img, 2 * H, H, WN, [&](int col, int k) {
            return (k < H) ? We1[(size_t)col * ld + k] : We1[(size_t)col * ld + H + 1 + (k - H)]; });
        for (int k = 0; k < H; ++k) { pk[g->be1 + k] = be1[k]; pk[g->w_er + k] = We1[(size_t)k * ld + H]; }
        const float* We2 = next((size_t)H * H); const float* be2 = next(H);
        pack_gemm_b(pk, g->we2_img, H, H, WN, [&](int col, int k) { return We2[(size_t)col * H + k]; });
        std::copy(be2, be2 + H, pk.begin() + g->be2);
    }
    {   // node_mlp.0 [H][2H]: columns [h | agg] (gcl.py:123-126)
        const float* Wn1 = next((size_t)H * 2 * H); const float* bn1 = next(H);
        pack_gemm_b(pk, g->wn1_img, 2 * H, H, WN, [&](int col, int k) { return Wn1[(size_t)col * 2 * H + k]; });
        if (g->node_split) pack_node_b_f32(pk, g->wn1_nimg, 2 * H, H, [&](int col, int k) { return Wn1[(size_t)col * 2 * H + k]; });
        std::copy(bn1, bn1 + H, pk.begin() + g->bn1);
        const float* Wn2 = next((size_t)H * H); const float* bn2 = next(H);
        pack_gemm_b(pk, g->wn2_img, H, H, WN, [&](int col, int k) { return Wn2[(size_t)col * H + k]; });
        if (g->node_split) pack_node_b_f32(pk, g->wn2_nimg, H, H, [&](int col, int k) { return Wn2[(size_t)col * H + k]; });
        std::copy(bn2, bn2 + H, pk.begin() + g->bn2);
    }
    if (c.coord_update) {
        const float* Wc1 = next((size_t)H * H); const float* bc1 = next(H); const float* wc2 = next(H);
        pack_gemm_b(pk, g->wc1_img, H, H, WN, [&](int col, int k) { return Wc1[(size_t)col * H + k]; });
        std::copy(bc1, bc1 + H, pk.begin() + g->bc1);
        std::copy(wc2, wc2 + H, pk.begin() + g->wc2);
    }
    if (c.attention) {
        const float* wa = next(H); const float* ba = next(1);
        std::copy(wa, wa + H, pk.begin() + g->wa);
        pk[g->ba] = ba[0];
    }
    if (p - src != n) return fail(HD_E_INVALID, "hd_egcl_set_weights: internal layout mismatch");
    if (g->dw_floats != pk.size()) {
        hipFree(g->dw); g->dw = nullptr;
        HD_TRY(dev_alloc(&g->dw, pk.size()));
        g->dw_floats = pk.size();
    }
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(g->dw, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice));
    g->weights_set = true;
    return HD_OK;
}

extern "C" int hd_egcl_graph_destroy(hd_egcl_graph* t) {
    if (!t) return HD_OK;
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    hipFree(t->row); hipFree(t->col); hipFree(t->cptr); hipFree(t->crows);
    hipFree(t->hin); hipFree(t->hres); hipFree(t->x4); hipFree(t->AB); hipFree(t->agg); hipFree(t->xagg); hipFree(t->Tn);
    hipFree(t->ea); hipFree(t->T1); hipFree(t->P); hipFree(t->M1); hipFree(t->C1); hipFree(t->geo); hipFree(t->trans); hipFree(t->ones);
    delete t;
    return HD_OK;
}

extern "C" int hd_egcl_graph_create(hd_egcl* g, const int* row, const int* col, int M, int E, hd_egcl_graph** out) {
    if (!g || !out || (E > 0 && (!row || !col))) return fail(HD_E_INVALID, "hd_egcl_graph_create: null argument");
    *out = nullptr;
    if (M < 1 || E < 0) return fail(HD_E_INVALID, "hd_egcl_graph_create: need M >= 1, E >= 0");
    for (int e = 0; e < E; ++e)
        if (row[e] < 0 || row[e] >= M || col[e] < 0 || col[e] >= M) return fail(HD_E_INVALID, "hd_egcl_graph_create: edge index out of range");
    HIP_TRY(hipSetDevice(g->device));
    std::vector<int> vr(row, row + E), vc(col, col + E), cptr(M + 1, 0), crows((size_t)E);
    for (int e = 0; e < E; ++e) cptr[vc[e] + 1]++;
    for (int i = 0; i < M; ++i) cptr[i + 1] += cptr[i];
    { std::vector<int> cur(cptr.begin(), cptr.end() - 1); for (int e = 0; e < E; ++e) crows[cur[vc[e]]++] = e; }
    hd_egcl_graph* t = new hd_egcl_graph();
    std::memset(t, 0, sizeof(*t));
    t->g = g; t->device = g->device; t->M = M; t->E = E;
    t->Mp = (M + 127) / 128 * 128; t->Ep = std::max(128, (E + 127) / 128 * 128);
    const int H = g->H;
    auto build = [&]() -> int {
        HD_TRY(dev_upload(&t->row, vr)); HD_TRY(dev_upload(&t->col, vc)); HD_TRY(dev_upload(&t->cptr, cptr)); HD_TRY(dev_upload(&t->crows, crows));
        auto zalloc = [&](float** p, size_t count) -> int {
            HD_TRY(dev_alloc(p, count));
            HIP_TRY(hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(float)));
            return HD_OK;
        };
        const size_t Mp = t->Mp, Ep = t->Ep;
        HD_TRY(zalloc(&t->hin, Mp * H)); HD_TRY(zalloc(&t->hres, Mp * H)); HD_TRY(zalloc(&t->x4, Mp * 4)); HD_TRY(zalloc(&t->AB, Mp * 2 * H));
        HD_TRY(zalloc(&t->agg, Mp * H)); HD_TRY(zalloc(&t->xagg, Mp * 4)); HD_TRY(zalloc(&t->Tn, Mp * H));
        HD_TRY(zalloc(&t->T1, Ep * H));         // (t->ea: the caller's edge_attr is read in place since round 3)
        HD_TRY(zalloc(&t->P, Ep * H)); HD_TRY(zalloc(&t->M1, Ep * H));
        HD_TRY(zalloc(&t->C1, Ep * H)); HD_TRY(zalloc(&t->geo, Ep * 4)); HD_TRY(zalloc(&t->trans, Ep * 4));
        std::vector<float> ones(Mp, 1.0f);
        HD_TRY(dev_upload(&t->ones, ones));
        return HD_OK;
    };
    const int r = build();
    if (r != HD_OK) { const std::string keep = g_err; hd_egcl_graph_destroy(t); g_err = keep; return r; }
    *out = t;
    return HD_OK;
}

static void egcl_gemm(hd_egcl* g, int epi, bool cat, const GemmArgs& a, hipStream_t s) {
    if (g->NS == 1) launch_gemm<4, 1, 1>(epi, cat, a, s);
    else launch_gemm<2, 2, 1>(epi, cat, a, s);
}

extern "C" int hd_egcl_forward(hd_egcl* g, hd_egcl_graph* t, const float* h, const float* x, const float* edge_attr,
                               const float* node_mask, const float* edge_mask, float* h_out, float* x_out,
                               float* edge_attr_out, void* stream) {
    if (!g || !t) return fail(HD_E_INVALID, "hd_egcl_forward: null handle/graph");
    if (t->g != g) return fail(HD_E_INVALID, "hd_egcl_forward: graph belongs to another handle");
    if (!g->weights_set) return fail(HD_E_STATE, "hd_egcl_forward: weights not set (hd_egcl_set_weights)");
    if (!h || !x || !h_out || !x_out) return fail(HD_E_INVALID, "hd_egcl_forward: null tensor");
    const hd_egcl_config& c = g->cfg;
    if (g->De > 0 && !edge_attr) return fail(HD_E_INVALID, "hd_egcl_forward: edge_attr required");
    if (c.edge_update && !edge_attr_out) return fail(HD_E_INVALID, "hd_egcl_forward: edge_attr_out required with edge_update");
    HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = (hipStream_t)stream;
    const int H = g->H, De = g->De, ctx = g->ctx, M = t->M, E = t->E;
    const float* W = g->dw;
    const bool wide = De == H;
    auto blocks = [](long long total) { return dim3((unsigned)((total + 255) / 256)); };
    auto gemm_args = [&](const float* A, int lda, int K1, int K, const float* A2, size_t img, size_t bias, float* Cc, int ldc,
                         int rows, int Nc, const float* nmask) {
        GemmArgs a;
        std::memset(&a, 0, sizeof(a));
        a.A = A; a.lda = lda; a.K1 = K1; a.K = K; a.A2 = A2; a.Bimg = W + img; a.bias = W + bias; a.C = Cc; a.ldc = ldc; a.M = rows;
        a.Nc = Nc; a.nmask = nmask;
        return a;
    };
    // Round 5, layers without context columns that update the coordinates (every layer of Edge_denoise): the caller's tensors are
    // read and written in place of the packed copies - no k_egcl_node_in / k_egcl_node_out launch (the residual comes from `h`, the
    // masked output goes straight to `h_out`, the coordinates leave with the sum over incoming edges) - and, for H-wide edge
    // attributes, k_egcl_pre's expression is the epilogue of the edge-attribute GEMM.  Same expressions, same order: same bits.
    const bool direct = ctx == 0 && c.coord_update && h_out != h && x_out != x;
    const float* hsrc = direct ? h : t->hin;
    const float* xsrc = direct ? x : t->x4;
    const int xs = direct ? 3 : 4;
    if (!direct) {   // node inputs
        EgclNodeInArgs a;
        a.h = h; a.x = x; a.hin = t->hin; a.hres = t->hres; a.x4 = t->x4; a.M = M; a.H = H; a.ctx = ctx;
        hipLaunchKernelGGL(k_egcl_node_in, blocks((long long)M * H), dim3(256), 0, s, a);
    }
    // Round 5, widths 128 / 256: the three node-level contractions (288 rows for a beam of 24: 20 workgroups of k_gemm, each a chain of
    // 8 - 16 K chunks behind barriers) run on k_node_split_f32 - 32 x 32 tiles, four K quarters per workgroup, weights requested at
    // entry: 9.3 / 14.4 / 9.0 us -> see profiles/r05_stage2_layer_kstats_node_split.log
    const bool nsplit = direct && g->node_split;
    auto nsargs = [&]() {
        NodeSplitArgs a;
        std::memset(&a, 0, sizeof(a));
        a.h_in = h; a.h_out = h_out; a.nmask = node_mask ? node_mask : t->ones; a.T = t->Tn; a.M = M; a.n_img = 1; a.norm = 1.0f;
        return a;
    };
    if (nsplit) {
        NodeSplitArgs a = nsargs();
        a.Wimg[0] = W + g->ab_nimg; a.bias[0] = W + g->ab_bias; a.ABout[0] = t->AB; a.upd = 0;
        if (H == 128) launch_node_split_f32_h<128, 3>(a, s); else launch_node_split_f32_h<256, 3>(a, s);
    } else {
        egcl_gemm(g, EPI_BIAS, false, gemm_args(hsrc, H, H, H, nullptr, g->ab_img, g->ab_bias, t->AB, 2 * H, M, 2 * H, nullptr), s);
    }
    if (E > 0) {
        if (wide && direct) {
            GemmArgs p = gemm_args(edge_attr, H, H, H, nullptr, g->w1e_img, g->zero_bias, t->P, H, E, H, nullptr);
            p.erow = t->row; p.ecol = t->col; p.ABn = t->AB; p.xn = xsrc; p.xs = xs; p.geo = t->geo; p.geo_mode = c.geo; p.colv = W + g->w_r;
            egcl_gemm(g, EPI_EGCL_PRE, false, p, s);
        } else if (wide) {
            egcl_gemm(g, EPI_BIAS, false, gemm_args(edge_attr, H, H, H, nullptr, g->w1e_img, g->zero_bias, t->T1, H, E, H, nullptr), s);
        }
        if (!(wide && direct)) {
            EgclPreArgs a;
            a.AB = t->AB; a.T1 = wide ? t->T1 : nullptr; a.ea = wide ? nullptr : edge_attr; a.w_e = W + g->w_e; a.w_r = W + g->w_r;
            a.w_c = W + g->w_c; a.hin = hsrc; a.x = xsrc; a.xs = xs; a.row = t->row; a.col = t->col; a.P = t->P; a.geo = t->geo;
            a.E = E; a.H = H; a.De = De; a.ctx = ctx; a.geo_mode = c.geo;
            hipLaunchKernelGGL(k_egcl_pre, blocks((long long)E * (H / 4)), dim3(256), 0, s, a);
        }
        egcl_gemm(g, EPI_BIAS_SILU, false, gemm_args(t->P, H, H, H, nullptr, g->w2_img, g->b2, t->M1, H, E, H, nullptr), s);
        {   // edge_feat = M (* att) * edge_mask, in place
            EgclRowArgs a;
            std::memset(&a, 0, sizeof(a));
            a.X = t->M1; a.w = W + g->wa; a.bias = W + g->ba; a.emask = edge_mask; a.E = E; a.H = H; a.attention = c.attention;
            hipLaunchKernelGGL((k_egcl_row<0>), dim3((E + 3) / 4), dim3(256), 0, s, a);
        }
        if (c.coord_update) {
            egcl_gemm(g, EPI_BIAS_SILU, false, gemm_args(t->M1, H, H, H, nullptr, g->wc1_img, g->bc1, t->C1, H, E, H, nullptr), s);
            EgclRowArgs a;
            std::memset(&a, 0, sizeof(a));
            a.X = t->C1; a.w = W + g->wc2; a.emask = edge_mask; a.geo = t->geo; a.trans = t->trans; a.range = c.coords_range;
            a.E = E; a.H = H; a.use_tanh = c.tanh;
            hipLaunchKernelGGL((k_egcl_row<1>), dim3((E + 3) / 4), dim3(256), 0, s, a);
        }
    }
    {   // sums over incoming edges (receiving index = col), ascending edge order
        CsrSumArgs cs;
        std::memset(&cs, 0, sizeof(cs));
        cs.ptr = t->cptr; cs.rows = t->crows; cs.M = M; cs.col0 = 0;
        cs.G = t->M1; cs.out = t->agg; cs.H = H; cs.ldo = H;
        cs.G2 = c.coord_update ? t->trans : nullptr; cs.out2 = t->xagg;      // the [E][4] translations in the same launch
        if (direct) { cs.x_in = x; cs.x_out = x_out; cs.xmask = node_mask; }   // ... and k_egcl_node_out's coordinate line with them
        hipLaunchKernelGGL(k_csr_sum, blocks((long long)M * (H / 4 + (c.coord_update ? 1 : 0))), dim3(256), 0, s, cs);
    }
    // node model: h_new = (h + node_mlp([h | agg])) (* node_mask)
    const float* nm = node_mask ? node_mask : t->ones;
    if (nsplit) {
        NodeSplitArgs p1 = nsargs(), p2 = nsargs();
        p1.Wimg[0] = W + g->wn1_nimg; p1.bias[0] = W + g->bn1; p1.agg_dense = t->agg;
        p2.Wimg[0] = W + g->wn2_nimg; p2.bias[0] = W + g->bn2; p2.resid_none = c.recurrent ? 0 : 1;
        if (H == 128) { launch_node_split_f32_h<128, 1>(p1, s); launch_node_split_f32_h<128, 2>(p2, s); }
        else { launch_node_split_f32_h<256, 1>(p1, s); launch_node_split_f32_h<256, 2>(p2, s); }
    } else {
    egcl_gemm(g, EPI_BIAS_SILU, true, gemm_args(hsrc, H, H, 2 * H, t->agg, g->wn1_img, g->bn1, t->Tn, H, M, H, nullptr), s);
    if (direct) {
        GemmArgs n2 = gemm_args(t->Tn, H, H, H, nullptr, g->wn2_img, g->bn2, h_out, H, M, H, nm);
        n2.resid = h; n2.ldr = H; n2.resid_none = c.recurrent ? 0 : 1;
        egcl_gemm(g, EPI_RESID_MASK, false, n2, s);
    } else {
        if (!c.recurrent) HIP_TRY(hipMemsetAsync(t->hres, 0, (size_t)t->Mp * H * sizeof(float), s));
        egcl_gemm(g, EPI_RESID_MASK, false, gemm_args(t->Tn, H, H, H, nullptr, g->wn2_img, g->bn2, t->hres, H, M, H, nm), s);
    }
    }
    if (c.edge_update && E > 0) {
        // edge_mlp: E1 = SiLU([edge_feat | edge_attr] We1^T + radial w_er + be1);  edge_attr' = (E1 We2^T + be2) * edge_mask
        // (round 3: the radial column + SiLU and the final mask ride in the GEMM epilogues, and the second GEMM writes the caller's
        // tensor - rows >= E are never stored: four stream operations less per layer, same expressions element by element)
        GemmArgs e1 = gemm_args(t->M1, H, H, 2 * H, edge_attr, g->we1_img, g->be1, t->C1, H, E, H, nullptr);
        e1.rowv = t->geo + 3; e1.rowv_stride = 4; e1.colv = W + g->w_er;
        egcl_gemm(g, EPI_RANK1_SILU, true, e1, s);
        egcl_gemm(g, edge_mask ? EPI_BIAS_MASK : EPI_BIAS, false,
                  gemm_args(t->C1, H, H, H, nullptr, g->we2_img, g->be2, edge_attr_out, H, E, H, edge_mask), s);
    }
    if (!direct) {
        EgclNodeOutArgs a;
        a.hnew = t->hres; a.hin = t->hin; a.x4 = t->x4; a.xagg = c.coord_update ? t->xagg : nullptr; a.nmask = node_mask;
        a.h_out = h_out; a.x_out = x_out; a.M = M; a.H = H; a.ctx = ctx;
        hipLaunchKernelGGL(k_egcl_node_out, blocks((long long)M * (H + ctx)), dim3(256), 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_linear(int device, const float* x, int M, int K, int ldx, const float* W, const float* b, int N, int act,
                         float* y, int ldy, void* stream) {
    if (!x || !W || !y) return fail(HD_E_INVALID, "hd_linear: null tensor");
    if (M < 0 || K < 1 || N < 1 || ldx < K || ldy < N || act < 0 || act > 2) return fail(HD_E_INVALID, "hd_linear: bad shape / act");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, "hd_linear: no such HIP device (is a GPU visible?)");
    if (M == 0) return HD_OK;
    HIP_TRY(hipSetDevice(device));
    LinArgs a;
    a.x = x; a.W = W; a.b = b; a.y = y; a.M = M; a.K = K; a.N = N; a.ldx = ldx; a.ldy = ldy; a.act = act;
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(k_linear, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

// ----------------------------------------------------------------------------- training GEMM (k_tgemm.hpp)

extern "C" int hd_gemm_f32(int device, int M, int N, int K, const float* A, long long a_m_stride, long long a_k_stride,
                           const float* B, long long b_k_stride, long long b_n_stride, float* C, int ldc, const float* bias,
                           int epi, const float* aux, const float* row_mask, float* C2, int split_k, float* ws,
                           float* colsum, void* stream) {
    if (!A || !B || !C) return fail(HD_E_INVALID, "hd_gemm_f32: null tensor");
    if (M < 0 || N < 0 || K < 0 || ldc < N) return fail(HD_E_INVALID, "hd_gemm_f32: bad shape");
    if ((a_m_stride != 1 && a_k_stride != 1) || (b_k_stride != 1 && b_n_stride != 1))
        return fail(HD_E_INVALID, "hd_gemm_f32: each operand needs one unit stride");
    if (epi < TG_EPI_BIAS || epi > TG_EPI_MUL_DSILU) return fail(HD_E_INVALID, "hd_gemm_f32: epi must be 0..3");
    if ((epi == TG_EPI_BIAS_SILU2 && !C2) || ((epi == TG_EPI_RESID_MASK || epi == TG_EPI_MUL_DSILU) && !aux))
        return fail(HD_E_INVALID, "hd_gemm_f32: the epilogue's second tensor is missing");
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && (!ws || epi != TG_EPI_BIAS)) return fail(HD_E_INVALID, "hd_gemm_f32: split-K needs a workspace and the plain epilogue");
    if (colsum && (a_m_stride != 1 || a_k_stride == 1)) return fail(HD_E_INVALID, "hd_gemm_f32: column sums ride on a GEMM with an m-contiguous A (dW = dY^T X)");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, "hd_gemm_f32: no such HIP device (is a GPU visible?)");
    if (M == 0 || N == 0) return HD_OK;
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const bool a_kc = a_k_stride == 1, b_kc = b_k_stride == 1;      // k-contiguous source: transposed on its way into LDS
    TGemmArgs g;
    std::memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.C2 = C2; g.bias = bias; g.aux = aux; g.rmask = row_mask;
    g.sam = a_m_stride; g.sak = a_k_stride; g.sbk = b_k_stride; g.sbn = b_n_stride;
    g.M = M; g.N = N; g.K = K; g.ldc = ldc; g.epi = epi;
    // split-K slabs are whole K chunks, so that a slab boundary never cuts a float4
    int kslab = std::max(K, 1);
    const bool splitting = split_k > 1;               // partial results through ws + k_tgemm_reduce, even if one slab is left
    if (splitting) {
        kslab = ((K + split_k - 1) / split_k + TG_BK - 1) / TG_BK * TG_BK;
        if (kslab < TG_BK) kslab = TG_BK;
        split_k = std::max(1, (K + kslab - 1) / kslab);
    }
    g.kslab = kslab;
    g.ws = splitting ? ws : nullptr;
    // (one slab: the n-tile-0 workgroups' sums over k ARE the column sums - no workspace, no reduce launch; short K, i.e. the node-level
    // dW of a small batch, where a training step is bound by the number of launches)
    g.colsum_ws = colsum ? (splitting ? ws + (size_t)split_k * M * N : colsum) : nullptr;
    auto aligned = [](const float* p, long long ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; };
    g.avec = aligned(A, a_kc ? a_m_stride : a_k_stride) ? 1 : 0;
    g.bvec = aligned(B, b_kc ? b_n_stride : b_k_stride) ? 1 : 0;
    dim3 grid((M + TG_BM - 1) / TG_BM, (N + TG_BN - 1) / TG_BN, split_k);
    const dim3 block(256);
    // a result of fewer than 128 tiles (a small training batch) as 32 x 32 tiles with the K quarters on the four wavefronts.  The
    // tile count bounds M * N below 1 M elements; the kernel re-reads a 32 x K strip of both operands per tile with no LDS reuse, so
    // K is bounded too: the measured family is the node-level Linears (K <= 2 hidden_nf = 512; profiles/r05_ab_small_gemm.log covers
    // it up to its upper end, M = 3,840 rows x N = 256 / 512: step 15.9 -> 15.3 ms at B = 128) - longer K takes the LDS-tiled kernel
    if (!splitting && (long long)grid.x * grid.y < 128 && K >= 32 && K <= 1024) {
        g.colsum = colsum;
        const dim3 sg((M + 31) / 32, (N + 31) / 32);
        if (a_kc && b_kc) hipLaunchKernelGGL((k_tgemm_small<true, true>), sg, block, 0, s, g);
        else if (a_kc) hipLaunchKernelGGL((k_tgemm_small<true, false>), sg, block, 0, s, g);
        else if (b_kc) hipLaunchKernelGGL((k_tgemm_small<false, true>), sg, block, 0, s, g);
        else hipLaunchKernelGGL((k_tgemm_small<false, false>), sg, block, 0, s, g);
        HIP_TRY(hipGetLastError());
        return HD_OK;
    }
    if (splitting) {                                   // 1-D, XCD-aware: the tiles of a slab share an XCD's L2 (k_tgemm.hpp)
        g.nx = grid.x; g.ny = grid.y; g.nz = split_k;
        grid = dim3(g.nx * g.ny * 8 * ((split_k + 7) / 8));
    }
    // FAST: full tiles, whole K chunks per slab, 16-byte aligned rows - no bounds checks, no branches around the loads
    const bool fast = g.avec && g.bvec && M % TG_BM == 0 && N % TG_BN == 0 && K % TG_BK == 0 && K > 0;
    auto launch = [&](auto Fast) {
        constexpr bool F = decltype(Fast)::value;
        if (a_kc && b_kc) hipLaunchKernelGGL((k_tgemm<true, true, F>), grid, block, 0, s, g);
        else if (a_kc) hipLaunchKernelGGL((k_tgemm<true, false, F>), grid, block, 0, s, g);
        else if (b_kc) hipLaunchKernelGGL((k_tgemm<false, true, F>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((k_tgemm<false, false, F>), grid, block, 0, s, g);
    };
    if (fast) launch(std::true_type{}); else launch(std::false_type{});
    if (splitting) {
        TGemmReduceArgs r;
        r.ws = ws; r.colsum_ws = g.colsum_ws; r.bias = bias; r.C = C; r.colsum = colsum; r.M = M; r.N = N; r.ldc = ldc; r.nz = split_k;
        const long long total = (long long)M * N + (colsum ? M : 0);
        hipLaunchKernelGGL(k_tgemm_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, r);
    }
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

// dW2 = G2^T P in fp16x3 arithmetic (k_dw2_f16): gmax / pmax = the n per-workgroup maxima the fp16x3 backward left in its workspace
extern "C" int hd_dw2_f16(int device, int rows, int H, const float* G2, const float* P, const float* gmax, const float* pmax, int n,
                          float* dW2, int ldc, float* ws, long long ws_floats, void* stream) {
    const char* who = "hd_dw2_f16";
    if (!G2 || !P || !dW2 || !ws || !gmax || !pmax || n < 1) return fail(HD_E_INVALID, std::string(who) + ": null tensor / no maxima");
    if (H != 128 && H != 256) return fail(HD_E_INVALID, std::string(who) + ": H must be 128 or 256 (narrower layers use hd_gemm_f32)");
    if (rows <= 0 || rows % 32 != 0 || ldc < H) return fail(HD_E_INVALID, std::string(who) + ": rows must be a positive multiple of 32 (whole edge tiles), ldc >= H");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, std::string(who) + ": no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    int slabs = (int)std::min<long long>(256, ws_floats / ((long long)H * H));
    slabs = std::max(1, std::min(slabs, rows / 128));
    if ((long long)slabs * H * H > ws_floats) return fail(HD_E_INVALID, std::string(who) + ": workspace smaller than one H x H slab");
    const int kslab = ((rows / 32 + slabs - 1) / slabs) * 32;
    slabs = (rows + kslab - 1) / kslab;
    Dw2F16Args a;
    a.G = G2; a.P = P; a.ws = ws; a.gmax = gmax; a.pmax = pmax; a.rows = rows; a.kslab = kslab; a.nmax = n;
    {
        static std::mutex mu;
        static unsigned long long prepared_mask = 0;
        std::lock_guard<std::mutex> lk(mu);
        const unsigned long long bit = 1ull << (device & 63);
        if (!(prepared_mask & bit)) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_dw2_f16<256>, hipFuncAttributeMaxDynamicSharedMemorySize, dw2_f16_lds_bytes<256>()));
            HIP_TRY(hipFuncSetAttribute((const void*)k_dw2_f16<128>, hipFuncAttributeMaxDynamicSharedMemorySize, dw2_f16_lds_bytes<128>()));
            prepared_mask |= bit;
        }
    }
    if (H == 256) hipLaunchKernelGGL((k_dw2_f16<256>), dim3(slabs), dim3(512), dw2_f16_lds_bytes<256>(), s, a);
    else hipLaunchKernelGGL((k_dw2_f16<128>), dim3(slabs), dim3(512), dw2_f16_lds_bytes<128>(), s, a);
    hipLaunchKernelGGL(k_dw2_reduce, dim3((H * H + 255) / 256), dim3(256), 0, s, ws, dW2, H * H, H, ldc, slabs);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

// ----------------------------------------------------------------------------- training loss (k_loss.hpp)
static int vlb_check(const char* who, int device, int B, int N, int D, int int_nf, int cont_nf) {
    if (B < 1 || N < 1 || D < 4 || int_nf < 0 || cont_nf < 0 || 3 + int_nf + cont_nf > D)
        return fail(HD_E_INVALID, std::string(who) + ": bad shape (B, N >= 1, D >= 4, 3 + int_nf + cont_nf <= D)");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, std::string(who) + ": no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    return HD_OK;
}

extern "C" int hd_vlb_loss_forward(int device, int B, int N, int D, int int_nf, int cont_nf, int l2_train, float T, float nv2, float nb2,
                                   float log_nv0, const float* net, const float* zt, const float* xh, const float* eps, const float* nm,
                                   const float* gam, const float* t_int, float* loss, float* err, void* stream) {
    HD_TRY(vlb_check("hd_vlb_loss_forward", device, B, N, D, int_nf, cont_nf));
    if (!net || !zt || !xh || !eps || !nm || !gam || !t_int || !loss || !err) return fail(HD_E_INVALID, "hd_vlb_loss_forward: null tensor");
    VlbArgs a;
    std::memset(&a, 0, sizeof(a));
    a.net = net; a.zt = zt; a.xh = xh; a.eps = eps; a.nm = nm; a.gam = gam; a.t_int = t_int; a.loss = loss; a.err = err;
    a.B = B; a.N = N; a.D = D; a.int_nf = int_nf; a.cont_nf = cont_nf; a.l2_train = l2_train; a.T = T; a.nv2 = nv2; a.nb2 = nb2; a.log_nv0 = log_nv0;
    hipLaunchKernelGGL((k_vlb<false>), dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_vlb_loss_backward(int device, int B, int N, int D, int int_nf, int cont_nf, int l2_train, float T, float nv2, float nb2,
                                    float log_nv0, const float* net, const float* zt, const float* xh, const float* eps, const float* nm,
                                    const float* gam, const float* t_int, const float* gout, float* dnet, float* dzt, float* dgam,
                                    void* stream) {
    HD_TRY(vlb_check("hd_vlb_loss_backward", device, B, N, D, int_nf, cont_nf));
    if (!net || !zt || !xh || !eps || !nm || !gam || !t_int || !gout || !dnet || !dzt || !dgam)
        return fail(HD_E_INVALID, "hd_vlb_loss_backward: null tensor");
    VlbArgs a;
    std::memset(&a, 0, sizeof(a));
    a.net = net; a.zt = zt; a.xh = xh; a.eps = eps; a.nm = nm; a.gam = gam; a.t_int = t_int; a.gout = gout; a.dnet = dnet; a.dzt = dzt; a.dgam = dgam;
    a.B = B; a.N = N; a.D = D; a.int_nf = int_nf; a.cont_nf = cont_nf; a.l2_train = l2_train; a.T = T; a.nv2 = nv2; a.nb2 = nb2; a.log_nv0 = log_nv0;
    hipLaunchKernelGGL((k_vlb<true>), dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

// z_t = alpha(g_t) xh + sigma(g_t) eps per molecule (dzt = NULL), or its gradient with respect to g_t (dzt given: dgt [B] written)
extern "C" int hd_vlb_zt(int device, int B, int ND, const float* xh, const float* eps, const float* gt, float* zt, const float* dzt,
                         float* dgt, void* stream) {
    if (B < 1 || ND < 1 || !xh || !eps || !gt || (!dzt && !zt) || (dzt && !dgt)) return fail(HD_E_INVALID, "hd_vlb_zt: bad argument");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, "hd_vlb_zt: no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    VlbZtArgs a{xh, eps, gt, zt, dzt, dgt, B, ND};
    if (dzt) hipLaunchKernelGGL((k_vlb_zt<true>), dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_vlb_zt<false>), dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

// W1 [H][2H + 2], b1 [H] -> Wst [2H][H], bst [2H], wrd [2][H] (dir 0); dWst, dwrd -> dW1 (dir 1: W1 is written, b1 / bst unused)
extern "C" int hd_edge_prep(int device, int H, int dir, float* W1, const float* b1, float* Wst, float* bst, float* wrd, void* stream) {
    if (H < 1 || !W1 || !Wst || !wrd || (dir == 0 && (!b1 || !bst))) return fail(HD_E_INVALID, "hd_edge_prep: bad argument");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, "hd_edge_prep: no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    EdgePrepArgs a{W1, b1, Wst, bst, wrd, H};
    const dim3 grid((H * (2 * H + 2) + 255) / 256);
    if (dir == 0) hipLaunchKernelGGL((k_edge_prep<0>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_edge_prep<1>), grid, dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

extern "C" int hd_colsum_f32(int device, int rows, int n, const float* const* src, const int* width, float* const* dst,
                             float* ws, void* stream) {
    if (n < 1 || n > 4 || !src || !width || !dst || !ws) return fail(HD_E_INVALID, "hd_colsum_f32: 1..4 arrays, a workspace");
    if (rows < 0) return fail(HD_E_INVALID, "hd_colsum_f32: rows < 0");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, "hd_colsum_f32: no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    ColSumArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rows = rows; a.n = n; a.ws = ws;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || width[i] < 1) return fail(HD_E_INVALID, "hd_colsum_f32: null array / width < 1");
        a.src[i] = src[i]; a.dst[i] = dst[i]; a.width[i] = width[i]; a.off[i + 1] = a.off[i] + width[i];
    }
    const int total = a.off[n];
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_colsum_stage1, dim3((total + 63) / 64, CS_CHUNKS), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_colsum_stage2, dim3((total + 255) / 256), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    return HD_OK;
}

// one pinned, device-visible result word per device (hipHostMalloc is mapped by default); calls are serialised by the lock, which is
// held for the few microseconds a call lasts
static std::mutex g_digest_mu;
static unsigned long long* g_digest_host[64] = {nullptr};

extern "C" int hd_params_digest(int device, const void* const* ptrs_dev, const long long* prefix_dev, int n, long long total,
                                unsigned long long* state_dev, unsigned long long* digest_host, void* stream) {
    if (!ptrs_dev || !prefix_dev || !state_dev || !digest_host) return fail(HD_E_INVALID, "hd_params_digest: null argument");
    if (n < 1 || total < 0) return fail(HD_E_INVALID, "hd_params_digest: n < 1 or total < 0");
    if (hd_device_count() <= device || device < 0 || device >= 64) return fail(HD_E_HIP, "hd_params_digest: no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    if (total == 0) { *digest_host = 0; return HD_OK; }
    std::lock_guard<std::mutex> lk(g_digest_mu);
    if (!g_digest_host[device]) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g_digest_host[device]), 64, hipHostMallocDefault));
    DigestArgs a;
    a.ptrs = reinterpret_cast<const uint32_t* const*>(ptrs_dev); a.prefix = prefix_dev; a.n = n; a.total = total;
    a.state = state_dev; a.host_out = g_digest_host[device];
    const long long groups = (total + DIGEST_CHUNK - 1) / DIGEST_CHUNK;
    hipLaunchKernelGGL(k_params_digest, dim3((unsigned)groups), dim3(256), 0, s, a);
    HIP_TRY(hipGetLastError());
    // the one host wait of the check, in stream order behind the writes the digest must see
    HIP_TRY(hipStreamSynchronize(s));
    *digest_host = *reinterpret_cast<volatile unsigned long long*>(g_digest_host[device]);
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
                     const uint32_t* draw_ptr, uint32_t draw0, hipStream_t s, const unsigned long long* base_ptr = nullptr) {
    ProfScope ps(h, s, 2);
    StepArgs a;
    a.zt = zt; a.eps = eps; a.coef = coef; a.nm = t->nm_bytes; a.zs = zs; a.noise = ns; a.draw_ptr = draw_ptr;
    a.step_ptr = step_ptr; a.base_ptr = base_ptr; a.draw0 = draw0; a.coef_rows = coef_rows; a.B = t->B; a.N = t->N; a.D = h->D; a.F = h->F;
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
    topo_use(topo, (hipStream_t)stream);
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
    topo_use(topo, s);
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
    topo_use(topo, s);
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
    h->sched_gen++;                            // captured graphs hold the old table addresses
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
    topo_use(topo, s);
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
    // hipGraph: one captured step whose step index / draw / time / sample base live in device memory, replayed
    // nsteps times.  The instantiated graph is kept with the topology and reused by later calls (it works on
    // library-owned copies of z and context, so nothing it has baked in moves between calls); it is rebuilt only when
    // something in GraphKey changes.  No host synchronisation anywhere: the caller's stream is ordered against the
    // replay stream with events.
    const size_t zbytes = (size_t)topo->B * topo->N * h->D * sizeof(float);
    const size_t cbytes = (size_t)topo->B * topo->N * h->cfg.context_node_nf * sizeof(float);
    hipStream_t rs = s;
    if (s == nullptr) {                                  // the legacy NULL stream cannot be captured
        if (!h->own_stream) HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        rs = h->own_stream;
        HIP_TRY(hipEventRecord(h->ev_in, s));
        HIP_TRY(hipStreamWaitEvent(rs, h->ev_in, 0));
    }
    GraphKey key;
    key.raw_x = raw_x; key.raw_h = raw_h; key.has_ctx = context ? 1 : 0; key.mol_shape = mol_shape < 0 ? -1 : mol;
    key.noise_rows = noise_rows; key.T = T; key.s_hi = raw_x ? s_hi : 0; key.seed = seed; key.weights_gen = h->weights_gen; key.sched_gen = h->sched_gen;
    // The replay state (d_step, d_draw, d_tcur, d_base) belongs to the handle: a replay issued on another stream - another
    // topology of this handle, or the same one from another caller stream - must have finished before this one touches it.
    if (h->ev_last_set) HIP_TRY(hipStreamWaitEvent(rs, h->ev_last, 0));
    if (topo->gexec && !(topo->gkey == key)) {
        if (h->ev_last_set) HIP_TRY(hipEventSynchronize(h->ev_last));   // a replay of the stale graph may still be running, on any stream
        HIP_TRY(hipStreamSynchronize(rs));
        hipGraphExecDestroy(topo->gexec);
        topo->gexec = nullptr;
    }
    if (!topo->gexec) {
        const int was_prof = h->prof;
        h->prof = 0;
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(rs, hipStreamCaptureModeThreadLocal));
        int rc = forward_impl(h, topo, topo->zbuf, h->d_tcur, 1, context ? topo->ctxbuf : nullptr, mol_shape < 0 ? -1 : mol,
                              topo->eps, rs);
        if (rc == HD_OK) {
            NoiseSrc ns = make_noise(raw_x, raw_h, noise_rows, seed, 0, draw0, share);
            rc = step_impl(h, topo, topo->zbuf, topo->eps, h->d_coef, 1, ns, mol, topo->zbuf, topo->N, h->d_step, h->d_draw,
                           draw0, rs, h->d_base);
        }
        if (rc == HD_OK) hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, rs, h->d_step, h->d_draw, h->d_tcur, h->d_tau);
        const hipError_t ce = hipStreamEndCapture(rs, &graph);
        h->prof = was_prof;
        if (rc != HD_OK) { if (graph) hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess) return fail(HD_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
        const hipError_t ie = hipGraphInstantiate(&topo->gexec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ie != hipSuccess) { topo->gexec = nullptr; return fail(HD_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
        topo->gkey = key;
    }
    HIP_TRY(hipMemcpyAsync(topo->zbuf, z, zbytes, hipMemcpyDeviceToDevice, rs));
    if (context) HIP_TRY(hipMemcpyAsync(topo->ctxbuf, context, cbytes, hipMemcpyDeviceToDevice, rs));
    hipLaunchKernelGGL(k_loop_state, dim3(1), dim3(1), 0, rs, h->d_step, h->d_draw, h->d_tcur, h->d_base, h->d_tau,
                       s_hi - 1, draw0, (unsigned long long)sample_id_base);
    for (int k = 0; k < nsteps; ++k) {
        const hipError_t le = hipGraphLaunch(topo->gexec, rs);
        if (le != hipSuccess) return fail(HD_E_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(le));
    }
    HIP_TRY(hipMemcpyAsync(z, topo->zbuf, zbytes, hipMemcpyDeviceToDevice, rs));
    HIP_TRY(hipEventRecord(h->ev_last, rs));
    h->ev_last_set = true;
    if (rs != s) {
        HIP_TRY(hipEventRecord(h->ev_out, rs));
        HIP_TRY(hipStreamWaitEvent(s, h->ev_out, 0));
    }
    return HD_OK;
}

// ----------------------------------------------------------------------------- measurement aid: sustained MFMA rate
extern "C" int hd_mfma_probe(int device, int kind, const float* in1024, float* scratch, int iters, double* ns_per_mfma_per_simd,
                             void* stream) {
    if (!in1024 || !scratch || !ns_per_mfma_per_simd) return fail(HD_E_INVALID, "hd_mfma_probe: null argument");
    if (kind < 0 || kind > 1 || iters < 1) return fail(HD_E_INVALID, "hd_mfma_probe: kind must be 0 (fp32) or 1 (fp16), iters >= 1");
    if (hd_device_count() <= device || device < 0) return fail(HD_E_HIP, "hd_mfma_probe: no such HIP device (is a GPU visible?)");
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const dim3 grid(2 * n_cu), block(256);                 // two wavefronts per SIMD: the edge kernels' occupancy
    auto launch = [&](int n) {
        if (kind == 0) hipLaunchKernelGGL((k_mfma_probe<0>), grid, block, 0, s, in1024, scratch, n);
        else hipLaunchKernelGGL((k_mfma_probe<1>), grid, block, 0, s, in1024, scratch, n);
    };
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    {
        const hipError_t ce = hipEventCreate(&e1);
        if (ce != hipSuccess) { (void)hipEventDestroy(e0); return fail(HD_E_HIP, std::string("hd_mfma_probe: ") + hipGetErrorString(ce)); }
    }
    launch(iters);                                          // warm-up (clocks, code)
    (void)hipEventRecord(e0, s);
    launch(iters);
    (void)hipEventRecord(e1, s);
    const hipError_t se = hipEventSynchronize(e1);
    float ms = 0.f;
    if (se == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (se != hipSuccess) return fail(HD_E_HIP, std::string("hd_mfma_probe: ") + hipGetErrorString(se));
    HIP_TRY(hipGetLastError());
    *ns_per_mfma_per_simd = (double)ms * 1e6 / ((double)iters * 8 * 2);     // 8 MFMAs per iteration and wavefront, 2 wavefronts per SIMD
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
