#!/bin/bash
# per-kernel averages of a 50-timestep sampler run (bench.py), top 14 kernels
export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python bench.py --timesteps 50 --steps 2 --warmup 1 --no-cpu-baseline "$@" > /tmp/b.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/ks/**/p_kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-52s calls %5d avg %8.1f us %5.1f%%" % (r["Name"][:52], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
