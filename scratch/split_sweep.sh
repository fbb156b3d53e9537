#!/bin/bash
# forward time at small batches with k_edge_split on (threshold off: every size) and off (HD_SPLIT_MAX_TILES=0), measurement
# build; usage: split_sweep.sh [precisions...]
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_batch or fp32_node_paths or forward_vs_oracle or forward_golden" 2>&1 | tail -4
for prec in ${@:-fp32 bf16x6 bf16x3}; do
for B in 1 2 8 16 18 24 32; do
  for th in 0 100000; do
    HIERDIFF_LIB=hierdiff_amd/lib/libhierdiff_hip_dbg.so HD_SPLIT_MAX_TILES=$th python scratch/time_fwd.py $prec $B 2>&1 | tail -1 | sed "s/^/split_max_tiles=$th  /"
  done
done
done
} > gpurun_out/split_sweep.log 2>&1
