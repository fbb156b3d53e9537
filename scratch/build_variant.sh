#!/bin/bash
# usage: scratch/build_variant.sh <name> [-DFLAG ...]  -> hierdiff_amd/lib/<name>.so (select with HIERDIFF_LIB=$PWD/hierdiff_amd/lib/<name>.so)
# the product's flags (hierdiff_amd/build.py) plus the given defines; spills are reported, not refused
name=$1; shift
cd "$(dirname "$0")/../hierdiff_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-value \
  -Wno-unused-result -Rpass-analysis=kernel-resource-usage "$@" -o ../lib/$name.so hierdiff_hip.hip 2> ../lib/$name.remarks
python3 - "$name" <<'PY'
import sys, re
name = sys.argv[1]
cur = None; bad = []
for line in open(f"../lib/{name}.remarks"):
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m: cur = m.group(1); continue
    m = re.search(r"(ScratchSize|VGPRs Spill)[^:]*: (\d+)", line)
    if m and int(m.group(2)) and cur: bad.append((cur, m.group(1), m.group(2)))
    if "error" in line: print(line.rstrip())
print(name, "spills:", bad if bad else "none")
PY
