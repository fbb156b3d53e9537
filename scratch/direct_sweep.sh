#!/bin/bash
# fp32 node GEMMs of small batches: k_gemm_direct (no LDS staging, 16 x 16 tiles) on / off; bitwise tests + forward times
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_batch or fp32_node_paths or forward_vs_oracle or forward_golden or general_edge" 2>&1 | tail -4
for B in 2 8 16 32 64 128 192; do
  for th in 0 100000; do
    HIERDIFF_LIB=hierdiff_amd/lib/libhierdiff_hip_dbg.so HD_DIRECT_MAX_ROWS=$th python scratch/time_fwd.py fp32 $B 2>&1 | tail -1 | sed "s/^/direct_max_rows=$th  /"
  done
done
} > gpurun_out/direct_sweep.log 2>&1
