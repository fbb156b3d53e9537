#!/bin/bash
# How much of the matrix-pipe time overlaps with VALU execution?  SQ_VALU_MFMA_COEXEC_CYCLES next to MFMA-busy / VALU-active,
# per precision mode (one --pmc pass each, kernel trace only).
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmc_coexec; rm -rf $OUT; mkdir -p $OUT
SET="SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES"
for prec in fp32 bf16x6 bf16x3; do
  rm -rf /tmp/pm
  timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pm -o p -- python bench.py --precision $prec --timesteps 3 --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-kernel-events > $OUT/run_$prec.log 2>&1
  f=$(find /tmp/pm -name "p_counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/$prec.csv
done
python - <<PY
import csv, glob, collections, os
for f in sorted(glob.glob("$OUT/*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if k.startswith("k_edge") or k.startswith("k_node<") or k.startswith("k_gemm"): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        m = {n: sum(v) / len(v) for n, v in d.items()}
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); co = m.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0); sq = m.get("SQ_BUSY_CYCLES", 1)
        print(f"{os.path.basename(f):12s} {k[:34]:34s} mfma_busy {busy / (32 * sq):.3f}  coexec/mfma_busy {co / busy if busy else 0:.3f}  "
              f"coexec/kernel {co / (32 * sq):.3f}  valu_active/wave_cycles {m.get('SQ_ACTIVE_INST_VALU', 0) / max(1, m.get('SQ_WAVE_CYCLES', 1)):.3f}  "
              f"insts valu {m.get('SQ_INSTS_VALU', 0):.0f} mfma {m.get('SQ_INSTS_MFMA', 0):.0f} trans {m.get('SQ_INSTS_VALU_TRANS_F32', 0):.0f}")
PY
