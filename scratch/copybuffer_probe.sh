#!/bin/bash
# Where do the __amd_rocclr_copyBuffer dispatches of the small-batch profiles come from?
# Kernel trace of scratch/time_fwd.py with 3 + R forwards for two values of R: if the copies belonged to the forward
# their count would grow with R; the trace also gives their position relative to the first forward kernel.
export TMPDIR=/tmp
for R in 20 60; do
  rm -rf /tmp/cb
  timeout 180 rocprofv3 --kernel-trace --output-format csv -d /tmp/cb -o p -- python scratch/time_fwd.py ${1:-fp32} ${2:-2} $R > /tmp/cb.log 2>&1
  python - "$R" <<'PY'
import csv, glob, sys
R = int(sys.argv[1])
rows = list(csv.DictReader(open(glob.glob('/tmp/cb/**/p_kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
cb = [i for i, n in enumerate(names) if 'copyBuffer' in n]
first_fwd = next(i for i, n in enumerate(names) if n.startswith('k_node_init'))
n_fwd = sum(1 for n in names if n.startswith('k_node_init'))
print(f"R={R}: {n_fwd} forwards, {len(cb)} copyBuffer dispatches, {sum(1 for i in cb if i < first_fwd)} of them before the first forward kernel "
      f"(dispatch index of the last copy {max(cb) if cb else -1}, of the first k_node_init {first_fwd})")
PY
  tail -1 /tmp/cb.log
done
