#!/bin/bash
# k_edge_mixed policy sweep (measurement build: python -m hierdiff_amd.build --debug-kernels):
#   HD_MIX_MAX_TILES=0 -> plain k_edge;  HD_MIX_ROUNDS=r -> at most r whole-tile rounds (0 = every tile column-split)
# usage: scratch/mix_sweep.sh "fp32 bf16x6" "40 48 64 80 96 112 128"
cd /root/repo
export HIERDIFF_LIB=/root/repo/hierdiff_amd/lib/libhierdiff_hip_dbg.so
for prec in ${1:-fp32}; do
  for B in ${2:-40 48 64 80 96 112 128}; do
    echo "== $prec B=$B"
    HD_MIX_MAX_TILES=0 python scratch/time_fwd.py $prec $B | sed 's/^/plain   /'
    for r in 0 1 2; do HD_MIX_ROUNDS=$r python scratch/time_fwd.py $prec $B | sed "s/^/rounds=$r /"; done
    python scratch/time_fwd.py $prec $B | sed 's/^/rule    /'
  done
done
