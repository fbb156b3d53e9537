#!/bin/bash
# same-box A/B of library builds on the edge kernels: scratch/ab_edge.sh "<prec batch> ..." reps libA libB ... -> gpurun_out/ab_edge.log
L=$PWD/hierdiff_amd/lib
out=gpurun_out/ab_edge.log; mkdir -p gpurun_out; : > $out
IFS=',' read -ra CFGS <<< "$1"; reps=$2; shift 2
for rep in $(seq 1 $reps); do
  for lib in "$@"; do
    for cfg in "${CFGS[@]}"; do
      echo "== $lib $cfg rep $rep" >> $out
      HIERDIFF_LIB=$L/$lib.so bash scratch/prof.sh $cfg 2>&1 | grep "k_edge\|ms/forward" >> $out
    done
  done
done
python3 - <<'PY'
import re, collections
d = collections.defaultdict(list)
cur = None
for line in open("gpurun_out/ab_edge.log"):
    if line.startswith("=="):
        p = line.split(); cur = (p[1], " ".join(p[2:-2])); continue
    m = re.search(r"(k_edge\w*<[^>]*>).*avg\s+([\d.]+) us", line)
    if m: d[(cur[1], m.group(1), cur[0])].append(float(m.group(2)))
    m = re.search(r"([\d.]+) ms/forward", line)
    if m: d[(cur[1], "ms/forward", cur[0])].append(float(m.group(1)))
print("---- means")
for k in sorted(d): print("%-12s %-34s %-18s mean %8.2f  %s" % (k[0], k[1], k[2], sum(d[k]) / len(d[k]), d[k]))
PY
