"""Experiment: two independent half-batches driven from two host threads on two HIP streams."""
import sys, time, threading, torch
sys.path.insert(0, '.')
from hierdiff_amd import DiffusionQM9, default_config
from hierdiff_amd.weights import synthetic_state_dict
DEV = "cuda:0"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 256
def build():
    m = DiffusionQM9(default_config(hidden_nf=256, n_layers=6, timesteps=T))
    sd = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.to(DEV); m.use_graph = False
    return m
models = [build() for _ in range(NS)]
streams = [torch.cuda.Stream(device=DEV) for _ in range(NS)]
masks = [torch.ones(B // NS, 30, 1, dtype=torch.bool, device=DEV) for _ in range(NS)]
def work(i):
    with torch.cuda.stream(streams[i]):
        models[i].sample_from_masks(masks[i], None, None, sample_id_base=i * (B // NS))
def run_all():
    th = [threading.Thread(target=work, args=(i,)) for i in range(NS)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
run_all()
t0 = time.perf_counter(); run_all(); run_all(); dt = (time.perf_counter() - t0) / 2
print(f"{NS} stream(s) x {B // NS} molecules: {dt / (T + 1) * 1e3:.3f} ms per forward-equivalent of {B} -> {B / (dt / (T + 1) * 1001):.1f} molecules/s")
