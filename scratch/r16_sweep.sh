#!/bin/bash
# fp32 forward times with the k_gemm_r16 node chain; HD_FUSE_MIN_ROWS (measurement build) moves the switch to the fused k_node_f32
cd /root/repo
export HIERDIFF_LIB=/root/repo/hierdiff_amd/lib/libhierdiff_hip_dbg.so
for B in 2 8 16 32 64 128 192; do python scratch/time_fwd.py fp32 $B 2>/dev/null; done
echo "== B=256: fused k_node_f32 (default) vs r16 chain"
python scratch/time_fwd.py fp32 256 2>/dev/null
HD_FUSE_MIN_ROWS=100000 python scratch/time_fwd.py fp32 256 2>/dev/null | sed 's/^/r16 chain /'
echo "== B=192, 128: fused instead of r16"
HD_FUSE_MIN_ROWS=0 python scratch/time_fwd.py fp32 192 2>/dev/null | sed 's/^/fused /'
HD_FUSE_MIN_ROWS=0 python scratch/time_fwd.py fp32 128 2>/dev/null | sed 's/^/fused /'
