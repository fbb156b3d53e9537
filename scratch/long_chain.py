"""Trajectory-level agreement over the workload's real length: T=1000 reverse-diffusion steps, injected normals,
HIP path vs the CPU oracle (small model so the oracle finishes in seconds)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_diffusion, DEV
from tests.helpers import rel_l2
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
from hierdiff_amd.noise_model import schedule_tables
H, L, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_list = [8, 5, 7, 3]
sd_np = synthetic_state_dict(9, 0, H, L, 2, True, 5, 1.0)
cfg = orc.DynCfg(in_node_nf=9, context_node_nf=0, hidden_nf=H, n_layers=L, normalization_factor=10.0)
nm, em = orc.canonical_masks(n_list)
B, N = nm.shape[:2]
g = torch.Generator().manual_seed(11)
raws = [(torch.randn(B, N, 3, generator=g), torch.randn(B, N, 8, generator=g)) for _ in range(T + 2)]
for prec in ("fp32", "bf16x3"):
    model = build_diffusion(sd_np, H, L, T=T, precision=prec)
    x, h = model.sample_from_masks(nm.to(DEV), em.to(DEV), None, raw_noises=raws)
    if prec == "fp32":
        # the oracle replays the gamma values the product evaluated (fp64 on the host, rounded)
        from hierdiff_amd.noise_model import evaluate_gamma
        import copy
        gg = evaluate_gamma(copy.deepcopy(model.gamma).cpu(), (torch.arange(T + 1, dtype=torch.float64) / T).view(-1, 1)).view(-1)
        t0 = time.time()
        xo, ho = orc.sample_chain(orc.as_torch_sd(sd_np), cfg, T, nm, em, None, raws, gamma_grid=gg)
        print(f"oracle chain {time.time() - t0:.1f} s")
    nmf = nm.float().numpy()
    print(prec, "x rel_l2", rel_l2(x.cpu().numpy() * nmf, xo.numpy() * nmf), "h rel_l2", rel_l2(h.cpu().numpy(), ho.numpy()),
          "max|dx|", float(np.abs(x.cpu().numpy() * nmf - xo.numpy() * nmf).max()))
