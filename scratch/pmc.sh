#!/bin/bash
# usage: scratch/pmc.sh "<counters...>" [tag]  -> SQ counters per kernel (time_fwd.py, bf16x3)
export TMPDIR=/tmp
cd /root/repo
rm -rf /tmp/pmc_out
timeout 240 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d /tmp/pmc_out -o p -- python scratch/time_fwd.py bf16x3 > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc_out -name "p_counter_collection.csv" | head -1)
python scratch/pmc_summary.py $f | grep -A40 "k_edge<256, false" | head -30
