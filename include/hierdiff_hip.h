/*
 * hierdiff_hip.h -- C ABI of libhierdiff_hip.so: the MI355X (gfx950) implementation of HierDiff's
 * coarse-grained reverse-diffusion hot path (EGNN dynamics forward + posterior step).
 *
 * Every entry point takes plain pointers and sizes; there are no torch / C++ types in the
 * signatures.  Device pointers are raw HIP device addresses (tensor.data_ptr()), `stream` is a
 * hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream), NULL = default stream.
 *
 * Conventions
 *   - return value 0 = OK, negative = error (HD_E_*); the message is in hd_last_error()
 *     (thread-local).  No exceptions or aborts cross the ABI.  Kernel faults surface at the
 *     caller's next synchronisation.
 *   - one handle per (device, stream of use); calls on one handle must be serialised by the
 *     caller; distinct handles are independent: the library keeps no process-global mutable
 *     state besides the thread-local hd_last_error() text.
 *   - the handle OWNS a repacked device copy of the weights; a topology OWNS its index tables
 *     and activation workspace; the caller owns every tensor it passes in.
 *
 * Reference interfaces replaced (file:line under /root/reference/endiffusion):
 *   hd_create / hd_set_weights   <- EGNN_dynamics_QM9.__init__ + load_state_dict
 *                                   (models/module/en_dynamics.py:9-36, sampler.py:27-34)
 *   hd_topology_create           <- get_adj_matrix + the mask tensors built in
 *                                   DiffusionQM9.sample (en_dynamics.py:124-143,
 *                                   train_module/diffusion_qm9.py:350-359)
 *   hd_egnn_forward              <- EGNN_dynamics_QM9._forward (en_dynamics.py:49-122), i.e.
 *                                   DiffusionQM9.phi (diffusion_qm9.py:135-138)
 *   hd_posterior_step            <- the arithmetic of sample_p_zs_given_zt after the network call
 *                                   (diffusion_qm9.py:328-345) incl. sample_normal (:438-456)
 *   hd_final_decode              <- sample_p_xh_given_z0 after the network call (:302-310)
 *   hd_noise                     <- sample_combined_position_feature_noise (:445-456)
 *   hd_sample_loop               <- the timestep loop of DiffusionQM9.sample (:375-384)
 */
#ifndef HIERDIFF_HIP_H
#define HIERDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HD_ABI_VERSION 12

#define HD_OK 0
#define HD_E_INVALID (-1)      /* bad argument / unsupported configuration */
#define HD_E_HIP (-2)          /* a HIP runtime call failed */
#define HD_E_NOMEM (-3)
#define HD_E_STATE (-4)        /* e.g. weights or schedule not set */

typedef struct hd_handle hd_handle;
typedef struct hd_topology hd_topology;

/* Mirrors EGNN_dynamics_QM9's constructor arguments (en_dynamics.py:9-13) that are on the path.
 * Unsupported values (mode != egnn_dynamics, sin_embedding, act_fn != silu) are rejected by the Python wrapper
 * before this struct is built. */
typedef struct hd_config {
    int32_t in_node_nf;          /* node features INCLUDING the time column, excluding context */
    int32_t context_node_nf;
    int32_t n_dims;              /* must be 3 */
    int32_t hidden_nf;           /* 32, 64, 128 or 256 (en_dynamics.py:9 accepts any width; production: 256).  Other multiples of 32
                                    are rejected by hd_create, not padded: every kernel is instantiated per width - the edge kernels
                                    keep H/32 accumulators in registers and deal H/32 column tiles to 4 or 8 wavefronts, the chunk
                                    images are cut into 1 KiB pieces per four wavefronts - and a zero-padded 256-wide network would
                                    change the summation order of every contraction, i.e. the bits, of a narrower model */
    int32_t n_layers;            /* number of EquivariantBlocks */
    int32_t inv_sublayers;       /* GCLs per block */
    int32_t attention;           /* 0/1 */
    int32_t tanh;                /* 0/1 */
    int32_t condition_time;      /* 0/1 */
    float norm_constant;
    float normalization_factor;
    float coords_range;          /* EGNN default 30; per-block range = coords_range / n_layers */
    int32_t precision;           /* matrix-core arithmetic of the H x H contractions:
                                    0 = exact fp32 (v_mfma_f32_32x32x2_f32) - what the reference computes in,
                                        and the default of the Python mirror,
                                    3 = "fp16x3" (opt-in): the per-edge contraction on a two-way FP16 split (11 + 11 significant
                                        bits per operand), 3 fp16 MFMAs per product, fp32 accumulation - truncation <= 2^-21 per
                                        product, at the rounding of the fp32 accumulation (measured 1.9e-7 rel-L2 on a 256-term
                                        contraction).  Operands are ranged by exact powers of two - W2 per matrix, the
                                        activations per edge row from a bound on the pre-activation known before the contraction
                                        starts - so FP16's exponent range imposes no assumption on the network.  The node GEMMs
                                        run the same arithmetic (hidden_nf >= 128; narrower: mode 0's node kernels).
                                    1 ("bf16x3") and 2 ("bf16x6") existed up to ABI 11 and are rejected since ABI 12: fp16x3
                                    is as accurate as 2 at the cost of 1 */
    int32_t aggregation_mean;    /* 0: aggregation_method 'sum' - neighbour sums / normalization_factor (egnn_new.py:280-282);
                                    1: 'mean' (:283-288) - sums / number of edge-list entries of the receiving node.  The
                                       reference's edge list holds all N x N pairs of a molecule, masked or not
                                       (en_dynamics.py:124-143), so that count is the padded N of the call for every node
                                       and normalization_factor is unused */
} hd_config;

int hd_version(void);
const char* hd_last_error(void);

/* Number of HIP devices visible (0 when there is no GPU; never fails). */
int hd_device_count(void);

int hd_create(const hd_config* cfg, int device, hd_handle** out);
int hd_destroy(hd_handle* h);

/* Number of fp32 values in the canonical weight blob: the dynamics parameters flattened in
 * state_dict registration order (hierdiff_amd/weights.py::dynamics_param_shapes). */
long long hd_weight_count(const hd_handle* h);

/* Load the canonical weight blob (host or device pointer). The library repacks it into its
 * kernel layouts; the source is not referenced after return. */
int hd_set_weights(hd_handle* h, const float* blob, long long n, int on_device, void* stream);

/* Build index tables for one (node_mask, edge_mask) pair.  Masks are HOST byte arrays
 * (0 = false): node_mask [B*N], edge_mask [B*N*N] row-major (b, i, j) or NULL for the canonical
 * mask node_mask[i] & node_mask[j] & (i != j).  The tables are laid out per molecule, so the bits
 * computed for a molecule do not depend on the rest of the batch (sharding a batch over ranks
 * reproduces the single-GPU result exactly). */
int hd_topology_create(hd_handle* h, const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N,
                       hd_topology** out);
/* The same, without a host wait: the tables travel pinned staging -> device in stream order of `stream` (the workspace
 * fill behind them), so the call returns as soon as the host-side layout is done and a training loop that meets new masks
 * every step keeps the GPU busy while the next batch's topology is laid out.  Launches on the topology from another stream
 * wait for the tables' arrival by themselves.  Device arena and staging buffer come from a grow-only pool that
 * hd_topology_destroy refills (no hipMalloc / hipFree / device-wide synchronisation per topology in steady state). */
int hd_topology_create_s(hd_handle* h, const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N, void* stream,
                         hd_topology** out);
/* Returns the topology's memory to the pool behind an event on the stream it was last used on (a topology that captured
 * a sampling graph or ran on several streams waits for the device instead). */
int hd_topology_destroy(hd_topology* t);
/* Frees every pooled arena (after their pending work). */
int hd_arena_pool_trim(void);
/* Host-only view of the edge-tile tables hd_topology_create builds for the same masks (no device needed): edges are
 * packed in 32-row tiles per molecule - cuts at molecule-relative multiples of 32, remainders of neighbouring
 * molecules share a tile at 4-row-aligned offsets - so the rows summed into one partial sum ("part") of a node
 * depend on its molecule alone.  counts5 = {active nodes, valid edges, tiles, parts, rows = 32 * tiles}; the
 * output arrays may be NULL: ei/ej [rows] receiving/sending compact node id, eseg [rows] segment of the row inside
 * its tile (255 = padding row), seg_part [rows] part id per (tile, segment), tile_nseg [tiles], pstart [nodes+1]
 * (a node's parts are pstart[i] .. pstart[i+1]-1, in the order consumers add them). */
int hd_topology_layout(const uint8_t* node_mask, const uint8_t* edge_mask, int B, int N, long long* counts5,
                       int* ei, int* ej, uint8_t* eseg, int* seg_part, int* tile_nseg, int* pstart);
/* info[0..5] = {B, N, active nodes, valid edges, edge tiles (32 edges), aggregation parts} */
int hd_topology_info(const hd_topology* t, long long* info6);

/* out[B,N,3+F] = EGNN_dynamics_QM9._forward(t, xh, node_mask, edge_mask, context, mol_shape).
 *   xh      device [B,N,3+F], F = in_node_nf - condition_time
 *   t       device, t_numel == 1 (broadcast) or B
 *   context device [B,N,context_node_nf] or NULL when context_node_nf == 0
 *   mol_shape  < 0 for None; otherwise nodes >= mol_shape keep their input coordinates
 * Stream-ordered; the NaN guard (whole-call reset of the velocity, en_dynamics.py:109-111) is
 * applied on the device without a host sync. */
int hd_egnn_forward(hd_handle* h, hd_topology* topo, const float* xh, const float* t, int t_numel,
                    const float* context, int mol_shape, float* out, void* stream);

/* Number of forwards since creation whose velocity contained NaN (syncs the stream). */
int hd_nan_events(hd_handle* h, void* stream, long long* count);

/* zs[B,mol,D] = posterior sample given the network output eps[B,N,D] (diffusion_qm9.py:326-345).
 *   coef   device [B,4] or [1,4] (coef_rows = B or 1): {alpha_t_given_s, sigma2_t_given_s,
 *          sigma_t, sigma = sigma_t_given_s * sigma_s / sigma_t}
 *   raw_x  device [noise_rows, mol, 3], raw_h device [noise_rows, mol, F]: the two randn draws;
 *          noise_rows = 1 reproduces fix_noise=True.  mol = mol_shape (< 0: N).
 *   zs may alias zt only when mol == N. */
int hd_posterior_step(hd_handle* h, hd_topology* topo, const float* zt, const float* eps, const float* coef,
                      int coef_rows, const float* raw_x, const float* raw_h, int noise_rows, int mol_shape,
                      float* zs, void* stream);

/* x[B,N,3], hfeat[B,N,F] = sample_p_xh_given_z0 after the network call.
 *   coef3 host {sigma_0, alpha_0, sigma_x}; noise as in hd_noise (raw normals or, with
 *   raw_x == NULL, the counter-based generator at (seed, sample_id_base + b, draw)). */
int hd_final_decode(hd_handle* h, hd_topology* topo, const float* z0, const float* eps, const float* coef3,
                    const float* raw_x, const float* raw_h, int noise_rows, uint64_t seed,
                    uint64_t sample_id_base, uint32_t draw, int share_rows, float* x, float* hfeat,
                    void* stream);

/* z[rows,N,3+F] = masked, centre-of-gravity-free combined noise from raw normals (rows = B), or,
 * with raw_x == NULL, from the library's counter-based generator (Philox4x32-10 + Box-Muller):
 * normal(seed, sample_id_base + b, draw, n*D + c).  share_rows != 0 draws one row (sample id
 * sample_id_base) and broadcasts it over the batch before masking (fix_noise). */
int hd_noise(hd_handle* h, hd_topology* topo, const float* raw_x, const float* raw_h, int noise_rows,
             uint64_t seed, uint64_t sample_id_base, uint32_t draw, int share_rows, float* z, void* stream);

/* Schedule for hd_sample_loop: host arrays of T+1 time values tau[k] = fp32(k)/T and T rows of
 * {alpha_t_given_s, sigma2_t_given_s, sigma_t, sigma} for s = 0..T-1 (t = s+1). */
int hd_set_schedule(hd_handle* h, int T, const float* tau, const float* coef4);

/* Runs posterior steps s = s_hi-1 ... s_lo on z[B,N,D] in place (rows >= mol_shape untouched):
 * per step one hd_egnn_forward at tau[s+1] and one hd_posterior_step.
 *   raw_x/raw_h  device [(s_hi-s_lo), noise_rows, mol, 3|F] in step order (first = s_hi-1), or NULL
 *                to use the counter-based generator with draw = T - s (draw 0 is z_T).
 *   use_graph    replay each step from a captured hipGraph (0 = plain launches).  The instantiated graph is
 *                cached with the topology and reused by later calls with the same arguments (any z / context /
 *                sample_id_base); it is stream-ordered like every other call - no host synchronisation. */
int hd_sample_loop(hd_handle* h, hd_topology* topo, float* z, const float* context, int mol_shape,
                   int s_hi, int s_lo, const float* raw_x, const float* raw_h, int noise_rows,
                   uint64_t seed, uint64_t sample_id_base, int use_graph, void* stream);

/* ---- Training primitives (the handle's hd_config.precision must be 0; the fp16x3 contractions are chosen per call below).
 * One "edge layer" is the part of a GCL / EquivariantUpdate that works on edges (egnn_new.py:35-56 / :91-104 with the
 * first Linear factorised): per unmasked edge (i, j)
 *     pre1 = A_i + B_j + |x_i - x_j|^2 w_r + |x0_i - x0_j|^2 w_d,   P = SiLU(pre1),   M = SiLU(W2 P + b2),
 *     GCL:   out_i = sum_j M sigmoid(wa.M + ba) / normalization_factor                          [M][H]
 *     COORD: out_i = sum_j u_ij tanh(wa.M) coords_range / normalization_factor  (xyz, 4th = 0)   [M][4]
 * with AB = [A | B] [M][2H] the node-level halves of the first Linear (computed by the caller, e.g. with a library
 * GEMM), x / x0 [M][4] the coordinates at block start / network input, M = active nodes, rows in the topology's
 * compact node order (hd_topology_nodes).  Weights are DEVICE pointers in state_dict layout: wrd [2][H] = the two
 * distance columns of the first Linear, W2 [H][H], b2 [H], wa [H], ba [1] (NULL: no attention bias).  The node-level Linears around an edge layer are plain GEMMs:
 * hierdiff_amd/training.py runs them on hd_gemm_f32 below (forward, dX and split-K dW; no BLAS-library kernel in a step). */
int hd_topology_nodes(const hd_topology* t, int* node_of /* host, `active nodes` ints: flat index b*N + n */);
/* The same order for a device consumer: `active nodes` int64 flat indices written to DEVICE memory in stream order. */
int hd_topology_nodes_device(hd_topology* t, long long* index, void* stream);
int hd_edge_layer_forward(hd_handle* h, hd_topology* topo, int coord, const float* AB, const float* x,
                          const float* x0, const float* wrd, const float* W2, const float* b2, const float* wa,
                          const float* ba, float* out, void* stream);
/* The same forward / backward with two per-call choices (hd_edge_layer_forward / _backward are the (precision 0, pre2 NULL) case).
 * (1) Keep the second-layer pre-activations instead of recomputing them: hd_edge_layer_save_rows = rows of a [rows][hidden_nf] fp32
 * buffer the forward of this topology can fill (the table's rows plus one spare tile; 0: the batch is small enough for the
 * column-split edge kernels, which keep their faster forward - pass pre2 = NULL and the backward recomputes).
 * hd_edge_layer_forward_s with pre2 != NULL writes W2 P + b2 of every edge row into it (accumulator order per 32-row tile, opaque to
 * the caller; 228 MB per layer at B = 256, N = 30, H = 256 - sized for this GPU's HBM, not for a 16 GB card);
 * hd_edge_layer_backward_s with the same buffer runs stage A as an element-wise kernel over it (no weight stream, no matrix
 * instruction).  Same results to the bit as the recomputing path (tests/test_gpu_training.py).
 * (2) precision: 0 = exact fp32; 3 = "fp16x3" (hidden_nf >= 128; narrower layers run the fp32 kernels): the sampler's two-way FP16
 * split in the training path - the reference trains with apex O2 (endiffusion/conf/trainer/default.yaml:4-5); here the forward
 * contraction (the fp16x3 edge kernel on the unscaled parameters; images, image scale and row ranges made on the device per call),
 * stage B's dP = G2 W2 (operand rows ranged by their exact maxima, which stage A leaves in f16ws) and dW2 (hd_dw2_f16) run on the
 * matrix cores proper, fp32-accurately, while everything around them stays exact fp32.  It exists only together with the kept pre2
 * (whose spare tile carries the image scalars from the forward to the backward call): hd_edge_layer_backward_s(precision 3) needs
 * the pre2 of a precision-3 forward and f16ws (hd_edge_layer_f16ws_floats floats, *n_wg = the number of per-workgroup maxima
 * hd_dw2_f16 reads at f16ws + 4 and f16ws + 4 + n_wg); where hd_edge_layer_save_rows(.., 3) is 0 the caller runs the layer in
 * precision 0.  (precision 2 = "bf16x6", the _p entry points and hd_dw2_x6 existed up to ABI 11.) */
long long hd_edge_layer_f16ws_floats(hd_handle* h, hd_topology* topo, int* n_wg);
int hd_dw2_f16(int device, int rows, int H, const float* G2, const float* P, const float* gmax, const float* pmax, int n,
               float* dW2, int ldc, float* ws, long long ws_floats, void* stream);
long long hd_edge_layer_save_rows(hd_handle* h, hd_topology* topo, int precision);
int hd_edge_layer_forward_s(hd_handle* h, hd_topology* topo, int coord, int precision, const float* AB, const float* x,
                            const float* x0, const float* wrd, const float* W2, const float* b2, const float* wa,
                            const float* ba, float* out, float* pre2, void* stream);
int hd_edge_layer_backward_s(hd_handle* h, hd_topology* topo, int coord, int precision, const float* AB, const float* x,
                             const float* x0, const float* wrd, const float* W2, const float* b2, const float* wa,
                             const float* ba, const float* gout, const float* pre2, float* f16ws, float* G2, float* P, float* G1, float* escal,
                             float* colpart, float* bapart, float* b2part, float* wrdpart, float* dAB, float* dx, float* dx0,
                             void* stream);
/* Backward of hd_edge_layer_forward given gout = dL/d(out).  Per-edge activations are recomputed; the caller provides
 * workspaces G2, P, G1 [rows][H], escal [rows][8], colpart, b2part [tiles][H], wrdpart [tiles][2][H], bapart [tiles]
 * (rows / tiles from hd_topology_layout's counts, tiles rounded up to a multiple of 4).  Written: dAB [M][2H],
 * dx, dx0 [M][4] and, for the caller's reductions over all edge rows,
 *     G2 = dL/d(W2 P + b2),  P                          =>  dW2 = G2^T P              (one dense GEMM, K = rows),
 *     per-tile partial sums                             =>  db2 = colsum(b2part),  d(wa) = colsum(colpart),
 *                                                           d(ba) = sum(bapart),
 *                                                           d(w_r), d(w_d) = colsum(wrdpart[:, 0]), colsum(wrdpart[:, 1]);
 * G1 = dL/d(pre1) is the operand of the two CSR sums behind dAB and is left in the workspace. */
int hd_edge_layer_backward(hd_handle* h, hd_topology* topo, int coord, const float* AB, const float* x,
                           const float* x0, const float* wrd, const float* W2, const float* b2, const float* wa,
                           const float* ba, const float* gout, float* G2, float* P, float* G1, float* escal, float* colpart,
                           float* bapart, float* b2part, float* wrdpart, float* dAB, float* dx, float* dx0, void* stream);

/* ---- Stage-2 layer: E_GCL forward (/root/reference/models/egnn/gcl.py:9-205), exact fp32.
 * The layer of the edge-denoise / refine models: messages from [h_row; h_col; radial; edge_attr; context], optional
 * attention gate, coordinate update and node update aggregated over the RECEIVING index `col`, optional update of the
 * H-wide edge features.  agg = 'sum', node_attr = None, act_fn = SiLU. */
typedef struct hd_egcl hd_egcl;
typedef struct hd_egcl_graph hd_egcl_graph;
typedef struct hd_egcl_config {        /* E_GCL.__init__ arguments (gcl.py:19) */
    int32_t hidden_nf;                 /* input_nf == output_nf == hidden_nf: 32, 64, 128 or 256 */
    int32_t edges_in_d;                /* hidden_nf (edge features) or < 32 (scalar edge attributes, e.g. 1) */
    int32_t context_nf;
    int32_t attention, tanh, coord_update, edge_update, recurrent;
    float coords_range;
    int32_t geo;                       /* 1: the message model sees 1 / radial^2 instead of radial (gcl.py:170-175); the coordinate and
                                          edge models keep radial.  An edge list with self edges then carries inf / NaN, as in the reference */
} hd_egcl_config;
int hd_egcl_create(const hd_egcl_config* cfg, int device, hd_egcl** out);
int hd_egcl_destroy(hd_egcl* g);
long long hd_egcl_weight_count(const hd_egcl* g);
/* Parameters flattened in state_dict registration order: mes_mlp.{0,2}, [edge_mlp.{0,2}], node_mlp.{0,2},
 * [coord_mlp.{0,2}], [att_mlp.0] (weight then bias each; coord_mlp.2 has no bias). */
int hd_egcl_set_weights(hd_egcl* g, const float* blob, long long n, int on_device, void* stream);
/* Edge list (HOST int32 arrays row, col [E], node ids < M) + workspaces; reusable across layers of the same width. */
int hd_egcl_graph_create(hd_egcl* g, const int* row, const int* col, int M, int E, hd_egcl_graph** out);
int hd_egcl_graph_destroy(hd_egcl_graph* t);
/* (h_out [M][H+ctx], x_out [M][3], edge_attr_out [E][H]) = E_GCL.forward(h [M][H+ctx], edges, x [M][3], edge_attr [E][De],
 * node_mask [M] or NULL, edge_mask [E] or NULL); all device fp32; edge_attr_out only with edge_update. */
int hd_egcl_forward(hd_egcl* g, hd_egcl_graph* t, const float* h, const float* x, const float* edge_attr,
                    const float* node_mask, const float* edge_mask, float* h_out, float* x_out, float* edge_attr_out,
                    void* stream);

/* y [M][ldy] (first N columns) = act(x [M][ldx] (first K columns) . W [N][K]^T + b [N] or NULL); device fp32, any M, K, N.
 * act: 0 none, 1 SiLU, 2 sigmoid.  The small dense layers around the E_GCL chains of the stage-2 model - torch.nn.Linear in
 * /root/reference/models/edge_denoise.py:29-33 (feature / edge / node embeddings) and :55-57 (focal / edge / node prediction
 * heads, Linear + SiLU + Linear [+ Sigmoid]).  One fmaf chain over k per output element. */
int hd_linear(int device, const float* x, int M, int K, int ldx, const float* W, const float* b, int N, int act,
              float* y, int ldy, void* stream);

/* General exact-fp32 GEMM of the training path (csrc/k_tgemm.hpp): C [M][ldc] = epi(sum_k A(m,k) B(k,n)) with
 * A(m,k) = A[m a_m_stride + k a_k_stride], B(k,n) = B[k b_k_stride + n b_n_stride], one unit stride per operand - the
 * node-level nn.Linear modules of the EGNN (egnn_new.py:17-33,76-89,172-173) under autograd: forward Y = X W^T + b,
 * backward dX = dY W and dW = dY^T X (diffusion_qm9.py:774-777 -> torch.autograd), and the dense reduction over all edge rows
 * dW2 = G2^T P of hd_edge_layer_backward.  epi: 0 C = acc + bias; 1 C = acc + bias, C2 = SiLU(C); 2 C = (aux + acc + bias) *
 * row_mask[m] (row_mask may be NULL); 3 C = acc * SiLU'(aux).  bias [N] or NULL.  split_k > 1 (epi 0 only): K is cut into
 * slabs whose partial results go to ws ([slabs][M][N] floats, + [slabs][M] when colsum is given) and are added in slab order
 * (deterministic); colsum [M] (any split_k, ABI 10: also 1 - no workspace then) receives sum_k A(m,k) (the bias gradient of
 * dW = dY^T X; A must be m-contiguous).  The slab count actually used is ceil(K / (32 * ceil(K / split_k / 32))) <= split_k. */
int hd_gemm_f32(int device, int M, int N, int K, const float* A, long long a_m_stride, long long a_k_stride,
                const float* B, long long b_k_stride, long long b_n_stride, float* C, int ldc, const float* bias,
                int epi, const float* aux, const float* row_mask, float* C2, int split_k, float* ws,
                float* colsum, void* stream);

/* (hd_dw2_f16, declared with the edge-layer functions above: dW2 [H][ldc] = G2^T P over `rows` edge rows - G2, P [rows][H] as
 * hd_edge_layer_backward_s leaves them, rows a multiple of 32, H = 128 or 256 - in fp16x3 arithmetic, csrc/k_dw2.hpp: the
 * fp32-accurate form of the one dense reduction over all edge rows.  ws: ws_floats >= H * H device floats; the product is cut into
 * min(256, ws_floats / H^2, rows / 128) slabs whose partial results are added in a fixed order: deterministic.) */
/* The variational training loss around the network call as one kernel per direction (round 5; reference: compute_loss with
 * t0_always = False, diffusion_qm9.py:530-673, and what it calls - compute_error :160-172, kl_prior :206-239, the log constants
 * :241-262, log_pxh_given_z0_without_constants :460-528).  All tensors fp32 on the device: net / zt / xh / eps [B][N][D] (network
 * output, noised input, normalised data [x | h], noise), nm [B][N], gam [4][B] = gamma at (s, t, 0, 1), t_int [B].  int_nf / cont_nf:
 * integer / continuous feature columns of the t = 0 likelihood (5 / 3 for node_coarse_type 'prop', 3 / 0 otherwise); l2_train: the
 * `l2` training loss; T timesteps; nv2 / nb2 = norm_values[2] / norm_biases[2]; log_nv0 = log(norm_values[0]).
 * forward: loss [B] (= nll per molecule), err [B] (= the `error` of the reference's info dict).
 * backward: given gout = dL/d(loss) [B]: dnet, dzt [B][N][D] and dgam [4][B] (the schedule network is trained).
 * hd_vlb_zt: z_t = sqrt(sigmoid(-g_t)) xh + sqrt(sigmoid(g_t)) eps (dzt = NULL), or dgt[b] = d/dg_t of that against dzt. */
int hd_vlb_loss_forward(int device, int B, int N, int D, int int_nf, int cont_nf, int l2_train, float T, float nv2, float nb2,
                        float log_nv0, const float* net, const float* zt, const float* xh, const float* eps, const float* nm,
                        const float* gam, const float* t_int, float* loss, float* err, void* stream);
int hd_vlb_loss_backward(int device, int B, int N, int D, int int_nf, int cont_nf, int l2_train, float T, float nv2, float nb2,
                         float log_nv0, const float* net, const float* zt, const float* xh, const float* eps, const float* nm,
                         const float* gam, const float* t_int, const float* gout, float* dnet, float* dzt, float* dgam, void* stream);
int hd_vlb_zt(int device, int B, int ND, const float* xh, const float* eps, const float* gt, float* zt, const float* dzt, float* dgt,
              void* stream);
/* Layout of the first edge Linear for the edge layer, one launch (round 5: a training step at the reference's batch size is bound
 * by the NUMBER of launches): W1 [H][2H + 2] (state_dict layout, columns h_row | h_col | radial | d0), b1 [H] -> Wst [2H][H] (the two
 * node halves stacked: AB = h Wst^T + bst), bst [2H] = [b1 | 0], wrd [2][H] (hd_edge_layer_forward's wrd).  dir 1: the way back for
 * the gradient - W1 receives dW1 assembled from Wst = dWst and wrd = dwrd (b1 / bst unused). */
int hd_edge_prep(int device, int H, int dir, float* W1, const float* b1, float* Wst, float* bst, float* wrd, void* stream);
/* Column sums of n <= 4 device arrays src[i] [rows][width[i]] into dst[i] [width[i]] in two launches, rows added in a fixed
 * order (32 ascending row ranges, then the ranges ascending): the reductions hd_edge_layer_backward leaves to its caller
 * (db2, d(wa), d(w_r) / d(w_d), d(ba) from the per-tile partial sums).  src / width / dst are HOST arrays of n entries;
 * ws: 32 * sum(width) device floats. */
int hd_colsum_f32(int device, int rows, int n, const float* const* src, const int* width, float* const* dst, float* ws,
                  void* stream);

/* 64-bit content digest of n device tensors of 32-bit words (the parameters a packed image was made from): the Python wrapper
 * compares it with the digest taken when it last called hd_set_weights / hd_set_schedule / hd_egcl_set_weights, so that a writer
 * which bumps no version counter (a fused optimizer, `.data` writes, an external kernel) can never leave the inference path on a
 * stale image.  The reference has no counterpart: it evaluates its modules in place (en_dynamics.py:49-122).
 * ptrs_dev [n] device pointers, prefix_dev [n + 1] word offsets (prefix_dev[n] == total), both DEVICE arrays; state_dev: two
 * device uint64 that are ZERO on entry (the kernel leaves them zero again).  One launch - the last workgroup publishes the sum to a
 * pinned host word - and one wait for `stream`.  The value is independent of the launch geometry. */
int hd_params_digest(int device, const void* const* ptrs_dev, const long long* prefix_dev, int n, long long total,
                     unsigned long long* state_dev, unsigned long long* digest_host, void* stream);

/* Host implementation of the library's normal generator (same bits as the device one up to libm
 * round-off); used by tests and by callers that want to reproduce a draw on the CPU. */
float hd_philox_normal_host(uint64_t seed, uint64_t sample_id, uint32_t draw, uint32_t index);

/* Kernel timing of the most recent hd_egnn_forward when profiling is enabled: per-kernel-family
 * accumulated milliseconds measured with HIP events on the caller's stream.
 * families: 0 edge (GCL+coord), 1 node GEMMs, 2 other.  `on` is a bitmask of the families to bracket
 * (1 = edge kernels only, 7 = all, 0 = off), optionally OR-ed with (stride << 8) to bracket only every
 * stride-th forward; enabling inserts event records only.  Launches replayed from a hipGraph are not
 * bracketed. */
int hd_profile_enable(hd_handle* h, int on);
int hd_profile_read(hd_handle* h, double* ms3, long long* launches3);

/* Measurement aid (no reference counterpart; bench.py's `roofline.sustained`): the rate at which THIS chip, under its power budget,
 * issues one matrix instruction when every SIMD streams it from register operands at the edge kernels' occupancy (two wavefronts per
 * SIMD, eight accumulators, operands taken from `in1024` - pass random data, zeros clock higher).  kind 0: v_mfma_f32_32x32x2_f32,
 * 1: v_mfma_f32_32x32x16_f16.  `scratch`: 2 * 256 * (number of CUs) device floats.  Runs the loop twice
 * (warm-up, timed with HIP events on `stream`) and waits for it.  *ns_per_mfma_per_simd = elapsed / (MFMAs issued per SIMD). */
int hd_mfma_probe(int device, int kind, const float* in1024, float* scratch, int iters, double* ns_per_mfma_per_simd, void* stream);

/* Debug aid (no reference counterpart), live only in a measurement build of the library
 * (python -m hierdiff_amd.build --debug-kernels; the product build returns 0): per-wave cycle stamps of the
 * handle's most recent traced edge-kernel launch (environment HD_ABLATE with bit 16 set at hd_create; H = 256,
 * GCL variant).  32 int64 per workgroup = 4 waves x {start|HW_ID<<48, loop start|XCC_ID<<48, loop end,
 * end, 3 epilogue stamps, segments}.  Returns the number of workgroups copied (0 if nothing was traced). */
int hd_debug_edge_trace(hd_handle* h, long long* out, int max_wg);

#ifdef __cplusplus
}
#endif
#endif /* HIERDIFF_HIP_H */
