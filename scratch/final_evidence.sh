#!/bin/bash
# End-of-round evidence on one box: GPU suite, smoke, rocprofv3 kernel stats + PMC passes (round_profiles.sh), default bench.
# usage: scratch/final_evidence.sh <tag> <suffix>   -> gpurun_out/{gpu_tests_<suffix>.log, smoke_<suffix>.log, prof_<tag>/, bench_default_<suffix>.json}
TAG=${1:-r03}; SFX=${2:-x}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$SFX.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_$SFX.log 2>&1
bash scratch/round_profiles.sh $TAG > gpurun_out/round_profiles_$SFX.log 2>&1
cp gpurun_out/prof_$TAG/${TAG}_counters.json profiles/${TAG}_counters.json
( time python bench.py ) > gpurun_out/bench_default_$SFX.json 2> gpurun_out/bench_default_$SFX.err
tail -3 gpurun_out/gpu_tests_$SFX.log; tail -1 gpurun_out/smoke_$SFX.log; tail -4 gpurun_out/bench_default_$SFX.err; head -c 600 gpurun_out/bench_default_$SFX.json
