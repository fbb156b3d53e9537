// Device kernels of libhierdiff_hip.so -- gfx950 (CDNA4) only.
//
// Design:
//   * the dense all-pairs edge list of the reference (en_dynamics.py:124-143) is never
//     materialised: only unmasked edges exist, packed in tiles of 32 edge rows;
//   * the first edge Linear is factorised, W1.[h_i;h_j;r;d0]+b = (W1a.h_i+b) + W1b.h_j + r.w_r + d0.w_d,
//     so per edge only an H x H contraction remains; it runs on the matrix cores, one 32-edge x H tile per
//     64-wide wavefront, either exactly in fp32 (v_mfma_f32_32x32x2_f32) or as three bf16 MFMAs per product
//     on head/tail-split fp32 operands ("bf16x3", v_mfma_f32_32x32x16_bf16; opt-in, the default is exact fp32);
//   * per-node sums over neighbours are wavefront-local, written as per-tile partial sums that the consuming
//     node kernel adds in a fixed order (bit-reproducible);
//   * in bf16x3 mode the whole row-local node chain (neighbour-sum reduction, node MLP, residual, the next
//     layers' first edge Linear) is one launch (k_node).
// Files: common.hpp (types, helpers, RNG), k_node.hpp, k_edge.hpp, k_edge_bwd.hpp (training: backward of an edge layer),
// k_sampling.hpp (output stage, posterior step, decode, noise), k_egcl.hpp (stage-2 layer E_GCL, forward).  (The slower one-wave-per-SIMD edge-kernel experiment of round 1 lives in scratch/experiments/.)
#pragma once
#include "common.hpp"
#include "k_node.hpp"
#include "k_edge.hpp"
#include "k_edge_bwd.hpp"
#include "k_sampling.hpp"
#include "k_egcl.hpp"
