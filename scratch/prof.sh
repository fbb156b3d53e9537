#!/bin/bash
# usage: scratch/prof.sh <precision> [batch]   -> per-kernel average durations of the dynamics forward
export TMPDIR=/tmp
rm -rf /tmp/prof_out
timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o p -- python scratch/time_fwd.py "$@" > /tmp/prof.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_out/**/p_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:9]:
    print(f"{r['Name'][:58]:58s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
tail -1 /tmp/prof.log
