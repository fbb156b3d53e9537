#!/bin/bash
cd /root/repo
export HIERDIFF_LIB=/root/repo/hierdiff_amd/lib/libhierdiff_hip_dbg.so
for B in 16 32 64 128 160; do
  HD_R16_SINGLE_ROWS=100000 python scratch/time_fwd.py fp32 $B 2>/dev/null | sed 's/^/RT=1 /'
  HD_R16_SINGLE_ROWS=0 python scratch/time_fwd.py fp32 $B 2>/dev/null | sed 's/^/RT=2 /'
done
HD_FUSE_MIN_ROWS=0 python scratch/time_fwd.py fp32 160 2>/dev/null | sed 's/^/fused /'
