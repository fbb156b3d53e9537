#!/bin/bash
# Round-6 evidence on ONE box and ONE library (gpurun_out/r06_final/...): GPU tier + smoke, kernel stats + PMC passes of both arithmetics
# (scratch/round_profiles.sh r06 -> r06_counters.json), default bench line (replays those counters: same library hash), sustained
# 20-step line, fuzz sweeps, training-step kernel tables, cost of the content digest.
# usage: scratch/r06_evidence.sh [suffix]
SFX=${1:-final}
OUT=gpurun_out/r06_$SFX; mkdir -p $OUT
export TMPDIR=/tmp
sha256sum hierdiff_amd/lib/libhierdiff_hip.so > $OUT/lib_sha256.txt
( time python -m pytest tests -m gpu -q ) > $OUT/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
bash scratch/round_profiles.sh r06 > $OUT/round_profiles.log 2>&1
cp gpurun_out/prof_r06/r06_counters.json profiles/r06_counters.json
cp gpurun_out/prof_r06/r06_counters.json gpurun_out/prof_r06/r06_*_T50_kernel_stats.csv gpurun_out/prof_r06/summary.txt $OUT/ 2>/dev/null
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 20 --warmup 2 --no-configs --no-cpu-baseline > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err
python tests/fuzz_parity.py 120 11 > $OUT/fuzz_parity.log 2>&1
python tests/fuzz_grads.py 24 13 big > $OUT/fuzz_grads_big.log 2>&1
python scratch/digest_cost.py > $OUT/digest_cost.log 2>&1
( bash scratch/train_kstats.sh 256 6 fp32; bash scratch/train_kstats.sh 256 6 fp16x3; bash scratch/train_kstats.sh 16 6 fp32 ) > $OUT/train_kstats.log 2>&1
tail -3 $OUT/gpu_tests.log; tail -1 $OUT/smoke.log; tail -3 $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json; echo; tail -2 $OUT/fuzz_parity.log; tail -2 $OUT/fuzz_grads_big.log
