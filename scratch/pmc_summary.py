"""Average PMC counter values per kernel from rocprofv3 counter_collection CSVs."""
import csv, collections, sys, glob
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(rows):
    if "k_edge" in k or "k_gemm" in k:
        print(k)
        for c, v in sorted(rows[k].items()):
            print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
