#!/bin/bash
# same-box A/B of the training step between two builds of the library: scratch/ab_lib_train.sh <libA> <libB> "<B list>" [precision]
# (names under hierdiff_amd/lib without .so; alternating runs, two repetitions)
A=$1; B=$2; BS=${3:-"16 64"}; TP=${4:-fp32}
for rep in 1 2; do
  for b in $BS; do
    for lib in $A $B; do
      echo -n "rep $rep lib $lib: "
      HIERDIFF_LIB=$PWD/hierdiff_amd/lib/$lib.so python scratch/train_step_time.py $b 6 $TP 2>&1 | tail -1 | cut -c1-120
    done
  done
done
