#!/bin/bash
# HBM traffic of the training kernels (round 5): FETCH_SIZE and WRITE_SIZE in their own passes over one fp16x3 (default) training-step
# run each, per kernel and launch, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request:
# hbm bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024), next to the algorithmic bytes of DESIGN.md section 4.  usage: pmc_train_hbm.sh [precision]
export TMPDIR=/tmp
TP=${1:-fp16x3}
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_th_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_th_$c -o p -- python scratch/train_step_time.py 256 6 $TP > /tmp/pmc_th_$c.log 2>&1
done
python - "$TP" <<'PY'
import csv, collections, glob, sys
def per_kernel(c):
    f = glob.glob(f'/tmp/pmc_th_{c}/**/p_counter_collection.csv', recursive=True)
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return d
fe, wr = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
E, H = 222720 + 0, 256          # edge rows of B = 256, N = 30 (the padded table has 6,960 tiles = 222,720 rows)
alg = {"k_edge_bwd<256, false, 0, 0, true>": 8 * H * E, "k_edge_bwd<256, true, 0, 0, true>": 8 * H * E,
       "k_edge_bwd<256, false, 1, 3, false>": 12 * H * E, "k_edge_bwd<256, true, 1, 3, false>": 12 * H * E,
       "k_edge_bwd<256, false, 1, 2, false>": 12 * H * E, "k_edge_bwd<256, false, 1, 0, false>": 12 * H * E,
       "k_dw2_f16<256>": 8 * H * E, "k_dw2_x6<256>": 8 * H * E, "k_csr_sum": 8 * H * E,
       "k_edge<256, false, 3, 768>": 4 * H * E, "k_edge<256, true, 3, 768>": 4 * H * E,
       "k_edge<256, false, 0, 256>": 4 * H * E, "k_edge<256, false, 2, 256>": 4 * H * E}
print(f"training step B=256 N=30 H=256 L=6, training_precision={sys.argv[1]}: HBM bytes per launch = (2 FETCH_SIZE + WRITE_SIZE) KiB * 1024")
for k in sorted(fe, key=lambda k: -sum(fe[k]) - sum(wr.get(k, [0]))):
    if not k.startswith("k_"): continue
    f, w = sum(fe[k]) / len(fe[k]), sum(wr[k]) / max(1, len(wr[k])) if k in wr else 0.0
    hbm = (2 * f + w) * 1024
    if hbm < 4e6: continue
    a = alg.get(k)
    print(f"{k:44s} launches {len(fe[k]):4d}  read {2*f*1024/1e6:8.1f} MB  written {w*1024/1e6:8.1f} MB  total {hbm/1e6:8.1f} MB" + (f"   algorithmic {a/1e6:7.1f} MB  ratio {hbm/a:.2f}" if a else ""))
PY
