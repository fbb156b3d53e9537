#!/bin/bash
# per-kernel averages of the dynamics forward at small batches (B = 64, 16, 2) in every precision,
# plus a kernel trace of one training step; writes gpurun_out/small_batch_prof.log
mkdir -p gpurun_out
{
for prec in fp32 fp16x3 bf16x6 bf16x3; do
  for B in 64 16 2; do
    echo "==== $prec B=$B"
    bash scratch/prof.sh $prec $B
  done
done
echo "==== training step B=256"
export TMPDIR=/tmp
rm -rf /tmp/tr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o p -- python scratch/train_step_time.py 256 > /tmp/tr.log 2>&1
tail -2 /tmp/tr.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/p_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.1f ms over the whole script (2 warm-up + 5 timed training steps + 5 no-grad forwards)" % (tot / 1e6))
for r in rows[:25]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
} > gpurun_out/small_batch_prof.log 2>&1
