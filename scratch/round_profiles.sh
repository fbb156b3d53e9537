#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/prof_r01 (copy into profiles/ afterwards):
#   kernel stats of a 50-timestep bench, and FETCH_SIZE / WRITE_SIZE counter passes of a 3-timestep bench.
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/prof_r01; rm -rf $OUT; mkdir -p $OUT
for prec in bf16x3 fp32; do
  rm -rf /tmp/ks
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python bench.py --precision $prec --timesteps 50 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_${prec}_T50.log 2>&1
  cp $(find /tmp/ks -name "p_kernel_stats.csv" | head -1) $OUT/r01_${prec}_T50_kernel_stats.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm -o p -- python bench.py --timesteps 3 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_pmc_$c.log 2>&1
  cp $(find /tmp/pm -name "p_counter_collection.csv" | head -1) $OUT/r01_bf16x3_pmc_${c}_counter_collection.csv
done
ls -la $OUT
