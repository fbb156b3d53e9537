"""Randomised sweep of the stage-2 layer E_GCL (hd_egcl_forward) against the CPU oracle: random graphs (dense per-molecule, sparse, with
repeated / self edges), widths, edge-attribute widths, context, attention / edge_update / coord_update / recurrent / tanh, masks present or
not, two chained layers.  usage: fuzz_stage2.py [cases] [seed]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from oracle import egnn_oracle as orc
from hierdiff_amd.stage2 import E_GCL, synthetic_egcl_state_dict
DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 13))
def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))
fails, worst = 0, 0.0
t0 = time.time()
for case in range(cases):
    H = int(rng.choice([32, 64, 128, 256]))
    wide = bool(rng.integers(0, 2))
    De = H if wide else int(rng.choice([1, 2, 4]))
    ctx = int(rng.choice([0, 0, 2]))
    att, eu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and wide
    cu, rec, tanh = bool(rng.integers(0, 2)), bool(rng.random() < 0.8), bool(rng.integers(0, 2))
    geo = bool(rng.random() < 0.25)
    M = int(rng.integers(2, 40))
    kind = int(rng.integers(0, 3))
    if kind == 0:                                   # dense blocks
        bs = int(rng.integers(1, 5)); n = max(2, M // bs); M = bs * n
        ar = torch.arange(n)
        row = ar.repeat_interleave(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n)
        col = ar.repeat(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n)
    if geo and kind != 1: kind = 1                  # 1 / radial^2: no self edges (inf in the reference too)
    if kind == 0:
        pass
    else:                                           # sparse random edges (kind 2: with repeats and self edges)
        E = int(rng.integers(1, 4 * M))
        row = torch.from_numpy(rng.integers(0, M, size=E)); col = torch.from_numpy(rng.integers(0, M, size=E))
        if kind == 1:
            keep = row != col
            if keep.sum() == 0:
                col[0] = (row[0] + 1) % M; keep[0] = True
            row, col = row[keep], col[keep]
    E = row.numel()
    masked = bool(rng.integers(0, 2)); has_em = bool(rng.integers(0, 2))
    nm = torch.from_numpy((rng.random((M, 1)) > 0.2).astype(np.float32)) if masked else None
    em = torch.from_numpy((rng.random((E, 1)) > 0.2).astype(np.float32)) if has_em else None
    h = torch.from_numpy(rng.standard_normal((M, H + ctx)).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((M, 3)).astype(np.float32))
    ea = torch.from_numpy(rng.standard_normal((E, De)).astype(np.float32))
    cfg = orc.EGCLCfg(hidden_nf=H, edges_in_d=De, context_nf=ctx, attention=att, tanh=tanh, coord_update=cu, edge_update=eu, recurrent=rec, geo=geo)
    layers = 2 if eu else 1
    hr, xr, er = h, x, ea
    hg, xg, eg = h.to(DEV), x.to(DEV), ea.to(DEV)
    line = f"case {case:3d} H={H:3d} De={De:3d} ctx={ctx} att={int(att)} eu={int(eu)} cu={int(cu)} rec={int(rec)} tanh={int(tanh)} geo={int(geo)} M={M:2d} E={E:4d} graph={kind} nm={int(masked)} em={int(has_em)}"
    try:
        for li in range(layers):
            sd_np = synthetic_egcl_state_dict(H, De, ctx, att, eu, 7000 + 10 * case + li, coord_gain=0.3)
            with torch.no_grad():
                out_r = orc.e_gcl_forward(orc.as_torch_sd(sd_np), cfg, hr, row, col, xr, er, nm, em)
            m = E_GCL(H, H, H, context_nf=ctx, edges_in_d=De, attention=att, tanh=tanh, coords_range=30, edge_update=eu,
                      coord_update=cu, recurrent=rec, geo=geo)
            own = set(m.state_dict().keys())            # (no coord_mlp without coord_update)
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items() if k in own})
            m = m.to(DEV)
            out_g = m(hg, [row.long().to(DEV), col.long().to(DEV)], xg, edge_attr=eg, node_mask=None if nm is None else nm.to(DEV),
                      edge_mask=None if em is None else em.to(DEV))
            hr, xr = out_r[0], out_r[1]
            hg, xg = out_g[0], out_g[1]
            if eu: er, eg = out_r[2], out_g[2]
        if not (torch.isfinite(hr).all() and torch.isfinite(xr).all()):
            # the REFERENCE arithmetic is non-finite here (geo: 1 / radial^2 of coincident points, e.g. two masked nodes that a
            # previous layer moved to the origin): the HIP layer must be non-finite in the same rows, nothing else is comparable
            r = 0.0
            bad = bool((torch.isfinite(hg.cpu()).all(dim=1) != torch.isfinite(hr).all(dim=1)).any())
            line += "  [reference non-finite]"
        else:
            r = max(rel(hg.cpu(), hr), rel(xg.cpu(), xr), rel(eg.cpu(), er) if eu else 0.0)
            bad = r > 1e-4 or not torch.isfinite(hg).all()
    except Exception as exc:                        # an unsupported combination must say so, not crash later
        r, bad = float("nan"), not isinstance(exc, NotImplementedError)
        line += f"  [{type(exc).__name__}: {str(exc)[:80]}]"
    worst = max(worst, 0.0 if r != r else r); fails += int(bad)
    print(line + f"  rel {r:.1e}{' FAIL' if bad else ''}", flush=True)
print(f"{cases} cases in {time.time() - t0:.0f} s, failures {fails}, worst rel-L2 {worst:.2e}")
sys.exit(1 if fails else 0)
