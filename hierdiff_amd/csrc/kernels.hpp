// Device kernels of libhierdiff_hip.so -- gfx950 (CDNA4) only.
//
// Design:
//   * the dense all-pairs edge list of the reference (en_dynamics.py:124-143) is never
//     materialised: only unmasked edges exist, packed in tiles of 32 edge rows;
//   * the first edge Linear is factorised, W1.[h_i;h_j;r;d0]+b = (W1a.h_i+b) + W1b.h_j + r.w_r + d0.w_d,
//     so per edge only an H x H contraction remains; it runs on the matrix cores, one 32-edge x H tile per
//     64-wide wavefront, either exactly in fp32 (v_mfma_f32_32x32x2_f32, the default) or, opt-in, on bf16 splits of the
//     fp32 operands with fp32 accumulation (v_mfma_f32_32x32x16_bf16): three pieces / six MFMAs per product ("bf16x6",
//     fp32-accurate) or two pieces / three MFMAs ("bf16x3");
//   * per-node sums over neighbours are wavefront-local, written as per-tile partial sums that the consuming
//     node kernel adds in a fixed order (bit-reproducible);
//   * the whole row-local node chain (neighbour-sum reduction, node MLP, residual, the next layers' first edge Linear) is
//     one launch in every mode (k_node for the bf16 splits, k_node_f32).
// Files: k_gemm_r16.hpp (fp32 node GEMMs of small / medium batches), common.hpp (types, helpers, RNG), k_node.hpp, k_edge.hpp, k_edge_split.hpp (fp32 edge kernel of very small batches), k_edge_bwd.hpp (training: backward of an edge layer),
// k_sampling.hpp (output stage, posterior step, decode, noise), k_egcl.hpp (stage-2 layer E_GCL, forward), k_tgemm.hpp (training: general fp32 GEMM of the node-level Linears, forward / dX / dW split-K), k_loss.hpp (training: the variational loss around the network call, one kernel per direction), k_digest.hpp (content digest of the parameter tensors: guards the packed weight images against silent staleness).  (The one-wave-per-SIMD edge-kernel experiments live in scratch/experiments/.)
#pragma once
#include "common.hpp"
#include "k_node.hpp"
#include "k_node_split.hpp"
#include "k_gemm_r16.hpp"
#include "k_edge.hpp"
#include "k_edge_split.hpp"
#include "k_edge_bwd.hpp"
#include "k_sampling.hpp"
#include "k_egcl.hpp"
#include "k_tgemm.hpp"
#include "k_dw2.hpp"
#include "k_loss.hpp"
#include "k_digest.hpp"
