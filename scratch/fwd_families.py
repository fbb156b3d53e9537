"""ms per forward and HIP-event averages of the edge / node / other kernel families for the headline batch, the mid-size BASELINE configs

averages of the edge / node kernel families for the headline batch, the mid-size BASELINE configs and the shipped B=2 batch."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd import _lib
from hierdiff_amd.weights import synthetic_state_dict
from hierdiff_amd.geom_stats import GEOM_FRAGMENT_HISTOGRAM as HIST

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sd_np = synthetic_state_dict(9, 0, 256, L, 2, True, 0, 1.0)
dyn = build_dynamics(sd_np, 256, L); dyn.precision = prec
lib = _lib.load()
rng = np.random.Generator(np.random.PCG64(2022))
keys = np.array([k for k in HIST if k <= 48]); p = np.array([HIST[k] for k in keys], float)
n3 = [int(v) for v in rng.choice(keys, size=256, p=p / p.sum())]
only = os.environ.get("FAM_CASES")
cases = [("B256_N30", [30] * 256, None), ("B64_N30", [30] * 64, None), ("geom256_pad48", n3, None), ("cfg5_B64_mol24", [30] * 64, "blk"),
         ("B128_N30", [30] * 128, None), ("B16_N30", [30] * 16, None), ("B2_N30", [30] * 2, None)]
if only:
    cases = [(f"B{b}_N30", [30] * int(b), None) for b in only.split(",")]
for name, sizes, em_kind in cases:
    B, N = len(sizes), max(max(sizes), 30 if em_kind else 0)
    N = 48 if name.startswith("geom") else N
    xh, nm, em = orc.random_inputs(sizes, 8, 1) if N == max(sizes) else (None, None, None)
    if xh is None:
        nm = (torch.arange(N)[None, :] < torch.tensor(sizes)[:, None]).unsqueeze(-1)
        g = torch.Generator().manual_seed(1)
        xh = torch.randn(B, N, 11, generator=g) * nm
    em_t = None
    if em_kind:
        e = torch.zeros(B, N, N, dtype=torch.bool); e[:, :24, :24] = True; e[:, 24:, 24:] = True
        em_t = (e & ~torch.eye(N, dtype=torch.bool)[None]).to(DEV)
    xh, nm = xh.to(DEV), nm.to(DEV)
    t = torch.full((B, 1), 0.5, device=DEV)
    topo = dyn.topology(nm, em_t, B, N); dyn.sync_weights()
    info = topo.info()
    for _ in range(5): o = dyn.forward_with_topology(topo, t, xh, None, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 30
    for _ in range(R): o = dyn.forward_with_topology(topo, t, xh, None, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / R
    _lib.check(lib.hd_profile_enable(dyn._handle(), 7), "pe")
    for _ in range(10): o = dyn.forward_with_topology(topo, t, xh, None, None)
    torch.cuda.synchronize()
    ms = (C.c_double * 3)(); cnt = (C.c_longlong * 3)()
    _lib.check(lib.hd_profile_read(dyn._handle(), ms, cnt), "pr")
    _lib.check(lib.hd_profile_enable(dyn._handle(), 0), "pe")
    fl = info["edges"] * (2.0 * 256 * 256 + 7 * 256)
    e_us = ms[0] / max(1, cnt[0]) * 1e3
    print(f"{name:16s} res={os.environ.get('HD_EDGE_RES','-')} {prec} tiles {info['tiles']:5d} rows {info['nodes']:5d}  {dt*1e3:7.3f} ms/fwd   edge {e_us:7.2f} us x{cnt[0]//10} "
          f"({fl / (e_us * 1e-6) / 1e12 / 157.3:.3f} of peak)  node {ms[1] / max(1, cnt[1]) * 1e3:6.2f} us x{cnt[1]//10}  other {ms[2] / max(1, cnt[2]) * 1e3:5.2f} us x{cnt[2]//10}  sum {o.double().abs().sum().item():.6e}", flush=True)
