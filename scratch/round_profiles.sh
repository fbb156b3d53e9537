#!/bin/bash
# Regenerates the rocprofv3 evidence of round 2 under gpurun_out/prof_r02 (copy into profiles/ afterwards):
#   per precision: kernel stats of a 50-timestep bench; FETCH_SIZE / WRITE_SIZE / SQ counter passes of a 3-timestep bench
#   (each --pmc set in its own run, with --kernel-trace only), then profiles/history/r02_counters.json via summarize_profiles.py.
# usage: scratch/round_profiles.sh [tag]     (tag defaults to r02)
export TMPDIR=/tmp
cd /root/repo
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for prec in ${PRECS:-fp32 fp16x3}; do
  rm -rf /tmp/ks
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python bench.py --precision $prec --timesteps 50 --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $OUT/bench_${prec}_T50.log 2>&1
  cp $(find /tmp/ks -name "p_kernel_stats.csv" | head -1) $OUT/${TAG}_${prec}_T50_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm -o p -- python bench.py --precision $prec --timesteps 3 --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-kernel-events > $OUT/bench_${prec}_pmc_$c.log 2>&1
    cp $(find /tmp/pm -name "p_counter_collection.csv" | head -1) $OUT/${TAG}_${prec}_pmc_${c}.csv
  done
  rm -rf /tmp/pm
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/pm -o p -- python bench.py --precision $prec --timesteps 3 --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-kernel-events > $OUT/bench_${prec}_sq.log 2>&1
  cp $(find /tmp/pm -name "p_counter_collection.csv" | head -1) $OUT/${TAG}_${prec}_sq.csv
  cp $(find /tmp/pm -name "p_kernel_trace.csv" | head -1) $OUT/${TAG}_${prec}_sq_kernel_trace.csv 2>/dev/null
done
python scratch/summarize_profiles.py $OUT $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
ls -la $OUT
