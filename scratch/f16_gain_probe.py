import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import _oracle_case, build_dynamics, DEV
from tests.helpers import rel_l2
from oracle import egnn_oracle as orc
torch.set_grad_enabled(False)
for n_list in ([30, 30, 17, 9], [30] * 40):
    for g in (1.0, 4.0, 10.0, 20.0, 40.0):
        sd_np, sd, cfg, xh, nm, em = _oracle_case(n_list, 256, 2, seed=505)
        for k in list(sd_np):
            if k.endswith("edge_mlp.0.weight") or k.endswith("edge_mlp.0.bias") or k.endswith("coord_mlp.0.weight") or k.endswith("coord_mlp.0.bias"):
                sd_np[k] = (sd_np[k] * g).astype(np.float32)
        B = xh.shape[0]
        t = torch.full((B, 1), 0.4)
        ref = orc.dynamics_forward(orc.as_torch_sd(sd_np), cfg, t, xh, nm, em, None, None, prefix="dynamics.egnn.").numpy()
        res = {}
        for precision in ("fp32", "bf16x3", "fp16x3"):
            dyn = build_dynamics(sd_np, 256, 2); dyn.precision = precision
            out = dyn._forward(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), None, None).cpu().numpy()
            res[precision] = (rel_l2(out[..., :3], ref[..., :3]), rel_l2(out[..., 3:], ref[..., 3:]))
        print(f"B={B} first-layer gain {g}: |ref| max {np.abs(ref).max():.3g}", {k: f"vel {v[0]:.1e} h {v[1]:.1e}" for k, v in res.items()})
