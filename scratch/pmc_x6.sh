#!/bin/bash
# SQ / LDS counters of the edge kernels in one precision mode (each --pmc set in its own run, kernel trace only)
export TMPDIR=/tmp
cd /root/repo
PREC=${1:-bf16x6}
OUT=gpurun_out/pmc_$PREC; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pm
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o p -- python bench.py --precision $PREC --timesteps 3 --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-kernel-events > $OUT/run$i.log 2>&1
  f=$(find /tmp/pm -name "p_counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/set$i.csv
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/set*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if k.startswith("k_edge"): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(f, k, {n: round(sum(v) / len(v), 1) for n, v in d.items()})
PY
