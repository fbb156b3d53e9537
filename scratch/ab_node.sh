#!/bin/bash
# same-box A/B of two library builds on the node-side kernels: scratch/ab_node.sh libA libB  -> gpurun_out/ab_node.log
L=$PWD/hierdiff_amd/lib
out=gpurun_out/ab_node.log; mkdir -p gpurun_out; : > $out
for rep in 1 2; do
  for lib in "$@"; do
    for cfg in "fp32 256" "bf16x6 256" "bf16x3 256"; do
      echo "== $lib $cfg rep $rep" >> $out
      HIERDIFF_LIB=$L/$lib.so bash scratch/prof.sh $cfg 2>&1 | grep "k_node\|r16\|ms/forward" >> $out
    done
  done
done
