#!/bin/bash
# same-box A/B of two library builds: alternate them N times, print the GCL edge-kernel average of each run
# usage: scratch/ab.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for l in $A $B; do
    printf "%-24s " $l
    HIERDIFF_LIB=$PWD/hierdiff_amd/lib/$l timeout 200 bash scratch/prof.sh bf16x3 | head -1 | awk '{print $(NF-2), $(NF-1)}'
  done
done
