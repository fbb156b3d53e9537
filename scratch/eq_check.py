import sys, torch, numpy as np
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV, rel_l2
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
H, L, B, N = 256, 6, 256, 30
sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 123, 1.0)
dyn = build_dynamics(sd_np, H, L)
xh, nm, em = orc.random_inputs([N] * B, 8, 9)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((B, 1), 0.4, device=DEV)
outs = [dyn._forward(t, xh, nm, em.to(DEV), None, None).cpu().numpy() for _ in range(4)]
for o in outs[1:]:
    d = np.abs(o - outs[0]); print("repeat diff max", d.max(), "rel", rel_l2(o, outs[0]), "n>1e-6", (d > 1e-6).sum())
g = torch.Generator().manual_seed(3)
q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g)); q = q.to(DEV)
xr = torch.cat([xh[..., :3] @ q, xh[..., 3:]], dim=-1)
outr = dyn._forward(t, xr, nm, em.to(DEV), None, None).cpu().numpy()
print("rot x", rel_l2((torch.from_numpy(outs[0][..., :3]).to(DEV) @ q).cpu().numpy(), outr[..., :3]), "h", rel_l2(outs[0][..., 3:], outr[..., 3:]))
perm = torch.randperm(N, generator=g).to(DEV)
outp = dyn._forward(t, xh[:, perm], nm, em.to(DEV), None, None).cpu().numpy()
print("perm", rel_l2(outs[0][:, perm.cpu().numpy()], outp))
