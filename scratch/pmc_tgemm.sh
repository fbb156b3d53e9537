#!/bin/bash
# SQ counters of the training GEMM (scratch/gemm_bench.py); separate passes per counter group; writes gpurun_out/pmc_tgemm.log
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
{
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc_out
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_out -o p -- python scratch/gemm_bench.py > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc_out -name "p_counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, collections, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "tgemm" in k:
        key = (k, r.get("Grid_Size", ""))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(rows[k].items())}, "n =", len(next(iter(rows[k].values()))))
PY
done
} > gpurun_out/pmc_tgemm.log 2>&1
