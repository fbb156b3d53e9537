import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import load, fixture_model, rel_l2
from oracle import egnn_oracle as orc
from hierdiff_amd.noise_model import step_coefficients
fx = load("f3_cond_h32_l2")
sd_np, sd, cfg = fixture_model(fx, context_node_nf=1)
s, t = torch.from_numpy(fx["s"]), torch.from_numpy(fx["t"])
gs, gt = orc.gamma_forward(sd, s), orc.gamma_forward(sd, t)
coef = step_coefficients(gs, gt).numpy()
z, eps, nm = fx["z"], fx["eps"], fx["node_mask"].astype(np.float32)
mol = int(fx["mol_shape"]); B = z.shape[0]
rx, rh = fx["raw_x"], fx["raw_h"]
out = np.zeros((B, mol, 11), np.float32)
for b in range(B):
    a, s2, st, sg = coef[b]
    ceps = (s2 / a) / st
    m = nm[b, :mol]            # [mol,1]
    cnt = m.sum()
    e = eps[b, :mol].copy()
    e[:, :3] -= (e[:, :3].sum(0) / cnt) * m
    n = np.concatenate([rx[0] * m, rh[0] * m], axis=1)
    n[:, :3] -= (n[:, :3].sum(0) / cnt) * m
    v = z[b, :mol] / a - ceps * e + sg * n
    v[:, :3] -= (v[:, :3].sum(0) / cnt) * m
    out[b] = v
print("emulated vs fixture", rel_l2(out, fx["zs"]), np.abs(out - fx["zs"]).max())
