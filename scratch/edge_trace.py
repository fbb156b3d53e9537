"""Per-wave cycle stamps of one k_edge launch (HD_ABLATE with bit 16): where does a wave-tile's time go?
Needs the measurement build of the library: python -m hierdiff_amd.build --debug-kernels (rebuild with --force afterwards)."""
import os, sys, ctypes, numpy as np, torch
os.environ.setdefault("HD_ABLATE", "16")
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
from hierdiff_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
PREC = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
sd_np = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs([30] * B, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((B, 1), 0.5, device=DEV)
dyn = build_dynamics(sd_np, 256, 6); dyn.precision = PREC
topo = dyn.topology(nm, None, B, 30); dyn.sync_weights()
for _ in range(5): o = dyn.forward_with_topology(topo, t, xh, None, None)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(32 * 4096, dtype=np.int64)
lib.hd_debug_edge_trace.restype = ctypes.c_int
n = lib.hd_debug_edge_trace(dyn._handle(), buf.ctypes.data_as(ctypes.c_void_p), 4096)
raw8 = buf[: n * 32].reshape(n, 4, 8).copy()         # [wg][wave][stamp]
raw = raw8[..., :4].copy()
hw = (raw[..., 0] >> 48) & 0xffff; xcc = (raw[..., 1] >> 48) & 0xf
tr = raw.copy(); tr[..., 0] &= (1 << 48) - 1; tr[..., 1] &= (1 << 48) - 1
pro = (tr[..., 1] - tr[..., 0]); loop = (tr[..., 2] - tr[..., 1]); epi = (tr[..., 3] - tr[..., 2])
print("workgroups", n)
for name, x in (("epi: silu+dot", raw8[..., 4] - raw8[..., 2]), ("epi: rowdot reduce", raw8[..., 5] - raw8[..., 4]), ("epi: att gather", raw8[..., 6] - raw8[..., 5]), ("epi: segment sums", raw8[..., 3] - raw8[..., 6])):
    print(f"{name:20s} mean {x.mean():9.0f}  p50 {np.percentile(x, 50):9.0f}")
print("mean segments per tile", raw8[..., 7].mean())
for name, x in (("prologue", pro), ("chunk loop", loop), ("epilogue", epi), ("wave total", tr[..., 3] - tr[..., 0])):
    print(f"{name:11s} mean {x.mean():9.0f}  p10 {np.percentile(x, 10):9.0f}  p50 {np.percentile(x, 50):9.0f}  p90 {np.percentile(x, 90):9.0f}  max {x.max():9.0f}")
# placement: (xcc, se, sh, cu) of wave 0 of each workgroup; which blocks share a CU, and when
cu = (hw[:, 0] >> 8) & 0xf; sh = (hw[:, 0] >> 12) & 1; se = (hw[:, 0] >> 13) & 7; simd = (hw[:, :] >> 4) & 3
key = xcc[:, 0] * 1000 + se * 100 + sh * 20 + cu
print("distinct CUs used:", len(set(key.tolist())), " simd ids of a workgroup's 4 waves (first 4 wgs):", simd[:4].tolist())
byc = {}
for b in range(n): byc.setdefault(int(key[b]), []).append(b)
some = list(byc.items())[:3]
for k, bl in some:
    st = [(b, int(tr[b, 0, 0] - tr[bl[0], 0, 0]), int(tr[b, 0, 3] - tr[bl[0], 0, 0])) for b in bl]
    print("CU", k, "blocks (id, start, end):", sorted(st, key=lambda z: z[1]))

# overlap of the two co-resident workgroups of a CU: for every wave, how much of its prologue / epilogue falls inside the
# chunk loop of the wave sharing its SIMD (matched by CU key + SIMD id)
import collections
waves = collections.defaultdict(list)
for b in range(n):
    for w in range(4):
        waves[(int(key[b]), int(simd[b, w]))].append((int(tr[b, w, 0]), int(tr[b, w, 1]), int(tr[b, w, 2]), int(tr[b, w, 3])))
cov_p, cov_e, alone = [], [], []
for lst in waves.values():
    for a in lst:
        def covered(lo, hi):
            c = 0
            for o in lst:
                if o is a: continue
                c += max(0, min(hi, o[2]) - max(lo, o[1]))
            return c / max(1, hi - lo)
        cov_p.append(covered(a[0], a[1])); cov_e.append(covered(a[2], a[3]))
        tot = 0
        for o in lst:
            if o is a: continue
            tot += max(0, min(a[3], o[3]) - max(a[0], o[0]))
        alone.append(1 - min(1.0, tot / max(1, a[3] - a[0])))
print(f"fraction of a wave's prologue covered by a SIMD-mate's chunk loop: mean {np.mean(cov_p):.2f}; epilogue: {np.mean(cov_e):.2f}; "
      f"fraction of its lifetime alone on the SIMD: {np.mean(alone):.2f}")
