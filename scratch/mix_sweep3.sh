#!/bin/bash
# ticket-queue form of k_edge_mixed: plain k_edge vs forced mixing (all whole rounds) vs the rule
cd /root/repo
export HIERDIFF_LIB=/root/repo/hierdiff_amd/lib/libhierdiff_hip_dbg.so
for prec in ${1:-fp32}; do
  for B in ${2:-40 64 96 128 160 192 256}; do
    echo "== $prec B=$B"
    HD_MIX_MAX_TILES=0 python scratch/time_fwd.py $prec $B 2>/dev/null | sed 's/^/plain  /'
    HD_MIX_ROUNDS=99 python scratch/time_fwd.py $prec $B 2>/dev/null | sed 's/^/mixed  /'
    python scratch/time_fwd.py $prec $B 2>/dev/null | sed 's/^/rule   /'
  done
done
