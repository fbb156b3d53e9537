#!/bin/bash
# SQ counters of the training kernels (k_edge_bwd stages, forward k_edge) in one training-step run -> gpurun_out/pmc_train.log
# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)  (numerator summed over 1,024 SIMDs, denominator over 32 SEs)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pmc_tr
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_tr -o p -- python scratch/train_step_time.py 256 > /tmp/pmc_tr.log 2>&1
python - <<'PY' > gpurun_out/pmc_train.log 2>&1
import csv, collections, glob
f = glob.glob('/tmp/pmc_tr/**/p_counter_collection.csv', recursive=True)
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    if "k_edge" in k:
        m = rows[k].get("SQ_VALU_MFMA_BUSY_CYCLES", []); b = rows[k].get("SQ_BUSY_CYCLES", [])
        if m and b:
            mm, bb = sum(m) / len(m), sum(b) / len(b)
            print(f"{k:40s} launches {len(m):4d}  MFMA_BUSY {mm:14.0f}  SQ_BUSY {bb:12.0f}  mfma_busy = {mm / (32 * bb):.3f}")
PY
cat gpurun_out/pmc_train.log
