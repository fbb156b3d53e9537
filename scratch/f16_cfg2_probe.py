import sys, copy, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_configs import _syn, build_diffusion, DEV
from hierdiff_amd import _lib
from oracle import egnn_oracle as orc
torch.set_grad_enabled(False)
H, L, T, B, N = 256, 9, 20, 8, 30
sd_np = _syn(H, L, seed=22, gain=0.02)
nm, em = orc.canonical_masks([N] * B)
g = torch.Generator().manual_seed(13)
raws = [(torch.randn(B, N, 3, generator=g), torch.randn(B, N, 8, generator=g)) for _ in range(T + 2)]
outs = {}
for prec in ("fp32", "fp16x3", "bf16x6"):
    model = build_diffusion(sd_np, H, L, T=T, precision=prec)
    x, h = model.sample_from_masks(nm.to(DEV), None, None, raw_noises=raws)
    cnt = C.c_longlong()
    _lib.check(_lib.load().hd_nan_events(model.dynamics._handle(), torch.cuda.current_stream().cuda_stream, C.byref(cnt)))
    outs[prec] = x.cpu()
    print(prec, "nan events", cnt.value, "max |x|", float(x.abs().max()))
for p in ("fp16x3", "bf16x6"):
    print(p, "vs fp32 rel", float((outs[p] - outs["fp32"]).norm() / outs["fp32"].norm()))
# single forwards along the fp32 trajectory: where do the modes part?
model32 = build_diffusion(sd_np, H, L, T=T, precision="fp32")
model16 = build_diffusion(sd_np, H, L, T=T, precision="fp16x3")
xh = torch.cat([raws[0][0], raws[0][1]], -1).to(DEV)
for scale in (1.0, 3.0, 10.0, 30.0):
    z = xh.clone(); z[..., :3] *= scale
    t = torch.full((B, 1), 0.5, device=DEV)
    a = model32.dynamics._forward(t, z, nm.to(DEV), em.to(DEV), None, None)
    b = model16.dynamics._forward(t, z, nm.to(DEV), em.to(DEV), None, None)
    print("coords x", scale, "fp16x3 vs fp32 forward rel", float((a - b).norm() / a.norm()), "vel", float((a[..., :3] - b[..., :3]).norm() / a[..., :3].norm()))
