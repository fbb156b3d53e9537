#!/bin/bash
# usage: scratch/ablate.sh <fp32|fp16x3> "<ablation values>"   (needs python -m hierdiff_amd.build --debug-kernels)
# HD_ABLATE bits (GCL edge kernel, H=256): 1 no epilogue, 2 no operand generation, 4 no barrier / W2 stream, 8 no AB gathers, 1024 no MFMA (round 6: the vector side alone)
PREC=${1:-fp32}; BATCH=${3:-256}
for abl in ${2:-0 1 4 8 12 13}; do
  printf "HD_ABLATE=%-3s " $abl
  HD_ABLATE=$abl HIERDIFF_LIB=$PWD/hierdiff_amd/lib/libhierdiff_hip_dbg.so timeout 300 bash scratch/prof.sh $PREC $BATCH | grep "k_edge<256, false" | head -1
done
