#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -S listing.
usage: count.py file.s <kernel-substring> [--blocks]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
show = "--blocks" in sys.argv
lines = open(path).read().splitlines()
# find kernel body
start = next(i for i, l in enumerate(lines) if re.match(r"^(_Z\S*%s\S*):" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_pk_"): return "vpk"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"
blocks = []; cur = ("entry", collections.Counter(), collections.Counter())
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith((";", "//", ".")) and not re.match(r"^\.LBB\S+:", s):
        continue
    m = re.match(r"^(\.LBB\S+):", s)
    if m:
        blocks.append(cur); cur = (m.group(1), collections.Counter(), collections.Counter()); continue
    op = s.split()[0]
    cur[1][cls(op)] += 1; cur[2][op] += 1
blocks.append(cur)
tot = collections.Counter(); ops = collections.Counter()
for name, c, o in blocks:
    tot += c; ops += o
    if show and sum(c.values()) > 8:
        print(f"{name:14s} n={sum(c.values()):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        if "--ops" in sys.argv:
            print("      " + " ".join(f"{k}:{v}" for k, v in o.most_common(14) if k.startswith("v_")))
print("TOTAL", dict(tot))
print("top valu:", " ".join(f"{k}:{v}" for k, v in ops.most_common(30) if k.startswith("v_")))
