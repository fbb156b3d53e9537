#!/bin/bash
# per-kernel totals of scratch/train_step_time.py (2 warm-up + 5 timed training steps + 5 no-grad loss values); usage: train_kstats.sh [B] [L] [fp32|fp16x3]
export TMPDIR=/tmp
rm -rf /tmp/tr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o p -- python scratch/train_step_time.py "$@" > /tmp/tr.log 2>&1
tail -1 /tmp/tr.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/p_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.1f ms over the whole script; %d kernel launches" % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:7.1f} ms {float(r['Percentage']):5.1f}%")
PY
