#!/bin/bash
# gradient-parity tests of the training path, then a kernel trace of the training step (B=256 and B=64);
# writes gpurun_out/train_prof.log
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -5
python scratch/train_step_time.py 256
python scratch/train_step_time.py 64
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
for r in rows[:30]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
} > gpurun_out/train_prof.log 2>&1
