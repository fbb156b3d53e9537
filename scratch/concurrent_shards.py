"""Does a small batch run faster as K independent sub-batches on K HIP streams (own handle + topology each)?  Molecules are
independent and the edge tiles are cut per molecule, so the results are bit-identical; the node kernels of one sub-batch
(few workgroups) can then overlap with the edge kernels of another.  usage: concurrent_shards.py [B] [T]"""
import sys, time, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
N, H, L, DEV = 30, 256, 6, torch.device("cuda:0")
sd = {k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 1.0).items()}
KMAX = 8
models = []
for _ in range(KMAX):
    m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L, timesteps=T)); m.load_state_dict(sd); models.append(m.to(DEV))
streams = [torch.cuda.Stream(DEV) for _ in range(KMAX)]
ref = {}
for prec in ("fp32", "bf16x6", "bf16x3"):
    for m in models: m.dynamics.precision = prec
    for use_graph in (True, False):
        for m in models: m.use_graph = use_graph
        for K in (1, 2, 4, 8):
            if B % K: continue
            b = B // K
            masks = [torch.ones(b, N, 1, dtype=torch.bool, device=DEV) for _ in range(K)]
            def run():
                outs = []
                for k in range(K):
                    with torch.cuda.stream(streams[k]):
                        outs.append(models[k].sample_from_masks(masks[k], None, None, sample_id_base=k * b))
                torch.cuda.synchronize(DEV)
                return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            x, h = run()
            t0 = time.perf_counter()
            for _ in range(2): x, h = run()
            dt = (time.perf_counter() - t0) / 2
            key = prec
            if K == 1 and use_graph: ref[key] = (x.clone(), h.clone())
            same = torch.equal(x, ref[key][0]) and torch.equal(h, ref[key][1])
            print(f"{prec:7s} graph={int(use_graph)} B={B} as {K} x {b}: {dt / (T + 1) * 1e3:.3f} ms per forward of the whole batch, "
                  f"{B / (dt / (T + 1) * 1001):.1f} molecules/s  bit-identical to K=1: {same}", flush=True)
