"""Gaps between consecutive MFMAs of the largest basic block of a kernel in a hipcc .s file: how many VALU / LDS / VMEM /
SALU instructions sit between them (a fp32 32x32x2 MFMA covers 16 VALU issue slots).  usage: mfma_gaps.py file.s mangled-name"""
import re, sys, collections
src, name = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ':'))
i1 = next(i for i in range(i0, len(lines)) if 's_endpgm' in lines[i])
blocks, cur = [], []
for l in lines[i0:i1]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'): continue
    if t.startswith('.LBB') or re.match(r's_c?branch', t):
        if cur: blocks.append(cur)
        cur = []
        continue
    cur.append(t.split()[0])
if cur: blocks.append(cur)
b = max(blocks, key=lambda x: sum(o.startswith('v_mfma') for o in x))
print('block instrs', len(b), 'mfma', sum(o.startswith('v_mfma') for o in b))
def cls(o):
    if o.startswith('v_mfma'): return 'M'
    if o.startswith('v_accvgpr'): return 'A'
    if o.startswith('v_'): return 'V'
    if o.startswith('ds_'): return 'D'
    if o.startswith('global_') or o.startswith('buffer_'): return 'G'
    if o.startswith('s_waitcnt'): return 'W'
    if o.startswith('s_barrier'): return 'B'
    if o.startswith('s_nop'): return 'N'
    return 'S'
seq = ''.join(cls(o) for o in b)
gaps = [len(g) for g in re.split('M', seq)]
vg = [sum(ch in 'VA' for ch in g) for g in re.split('M', seq)]
h = collections.Counter(min(v, 40) // 4 * 4 for v in vg)
print('VALU ops between consecutive MFMAs (bucket: count):', sorted(h.items()))
print('total VALU', seq.count('V'), 'accvgpr', seq.count('A'), 'ds', seq.count('D'), 'vmem', seq.count('G'), 'waitcnt', seq.count('W'), 'nop', seq.count('N'))
big = [(i, v) for i, v in enumerate(vg) if v > 16]
print('gaps > 16 VALU:', len(big), 'sum of excess', sum(v - 16 for _, v in big), big[:40])
if len(sys.argv) > 3:
    print(seq)
