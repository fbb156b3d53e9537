import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import load, fixture_model, rel_l2
from tests.test_gpu_parity import build_diffusion, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd.noise_model import step_coefficients
fx = load("f3_cond_h32_l2")
sd_np, sd, cfg = fixture_model(fx, context_node_nf=1)
model = build_diffusion(sd_np, 32, 2, C_=1)
z, nm, em, ctx = (torch.from_numpy(fx[k]) for k in ("z", "node_mask", "edge_mask", "context"))
s, t = torch.from_numpy(fx["s"]), torch.from_numpy(fx["t"])
mol = int(fx["mol_shape"])
g = model._gamma_host()
gs, gt = g(s), g(t)
ogs, ogt = orc.gamma_forward(sd, s), orc.gamma_forward(sd, t)
print("gamma diff", (gs-ogs).abs().max().item(), (gt-ogt).abs().max().item())
coef = step_coefficients(gs, gt)
print("coef", coef[0])
eps = model.phi(z.to(DEV), t.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV), mol).cpu()
print("eps rel", rel_l2(eps.numpy(), fx["eps"]))
raw = (torch.from_numpy(fx["raw_x"]), torch.from_numpy(fx["raw_h"]))
zs = model.sample_p_zs_given_zt(s.to(DEV), t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV), fix_noise=True, mol_shape=mol, raw_noise=raw).cpu()
ref = fx["zs"]
d = zs.numpy() - ref
print("zs rel", rel_l2(zs.numpy(), ref))
print("per-col err", np.abs(d).max(axis=(0,1)))
print("per-batch err", np.abs(d).max(axis=(1,2)))
print("per-node err", np.abs(d).max(axis=(0,2)))
print(nm[:, :, 0].int())
# regress the error on the three terms (features only, valid nodes)
m = nm[:, :mol, 0].bool().numpy()
zz = z[:, :mol, 3:].numpy()[m].reshape(-1)
ee = fx["eps"][:, :mol, 3:][m].reshape(-1)
nn_ = np.broadcast_to(fx["raw_h"], (4, mol, 8))[m].reshape(-1)
dd = d[:, :, 3:][m].reshape(-1)
A = np.stack([zz, ee, nn_], 1)
coefs, res, *_ = np.linalg.lstsq(A, dd, rcond=None)
print("lstsq coef on [z, eps, noise]:", coefs, "resid", np.abs(A @ coefs - dd).max())
print("---- manual call")
from hierdiff_amd import _lib
from hierdiff_amd.dynamics import _stream
g = model._gamma_host()
sD, tD = s.to(DEV), t.to(DEV)
coef2 = step_coefficients(g(sD.detach().float().cpu()), g(tD.detach().float().cpu()))
print("coef2 == coef:", torch.equal(coef2, coef), coef2[0].tolist())
coefD = coef2.to(DEV)
print("coefD back:", coefD.cpu()[0].tolist(), coefD.dtype, coefD.is_contiguous(), coefD.shape)
zt_c = z.to(DEV).contiguous()
epsD = torch.from_numpy(fx["eps"]).to(DEV).contiguous()
rx, rh = raw[0].to(DEV).contiguous(), raw[1].to(DEV).contiguous()
topo = model.dynamics.topology(nm.to(DEV), em.to(DEV), 4, 8)
zs2 = torch.empty((4, mol, 11), device=DEV)
_lib.check(_lib.load().hd_posterior_step(model._lib_handle(), topo.ptr, zt_c.data_ptr(), epsD.data_ptr(), coefD.data_ptr(), 4,
     rx.data_ptr(), rh.data_ptr(), 1, mol, zs2.data_ptr(), _stream(torch.device(DEV))))
torch.cuda.synchronize()
print("manual zs rel", rel_l2(zs2.cpu().numpy(), ref))
# same but with detached coef
coefD2 = coef2.detach().clone().to(DEV)
_lib.check(_lib.load().hd_posterior_step(model._lib_handle(), topo.ptr, zt_c.data_ptr(), epsD.data_ptr(), coefD2.data_ptr(), 4,
     rx.data_ptr(), rh.data_ptr(), 1, mol, zs2.data_ptr(), _stream(torch.device(DEV))))
torch.cuda.synchronize()
print("manual zs rel (detached coef)", rel_l2(zs2.cpu().numpy(), ref))
