"""How reproducible is the reference's fp32 GammaNetwork across hosts / batch shapes?  (DESIGN.md section 2 finding.)
Evaluates the schedule network of the F4 fixture (synthetic weights, seed 0) in fp32 on THIS host - as a [1001,1] batch,
row by row, and with different thread counts - and compares with the table the reference produced in the build
container (tests/golden/f4_schedule.npz) and with the fp64 evaluation the product uses."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
from hierdiff_amd.noise_model import GammaNetwork, evaluate_gamma
fx = dict(np.load("tests/golden/f4_schedule.npz"))
sd = orc.as_torch_sd(synthetic_state_dict(9, 0, 32, 1, 2, True, int(fx["weight_seed"])))
T = int(fx["T"])
tau = (torch.arange(T + 1, dtype=torch.int64).view(-1, 1) / T)
net = GammaNetwork(); net.load_state_dict({k[6:]: v for k, v in sd.items() if k.startswith("gamma.")})
g64 = evaluate_gamma(net, tau).view(-1).numpy()
def s2(g):
    g = torch.from_numpy(np.asarray(g, np.float64))
    return (-torch.expm1(torch.nn.functional.softplus(g[:-1]) - torch.nn.functional.softplus(g[1:]))).numpy()
print(f"host: {open('/proc/cpuinfo').read().split('model name')[1].split(':')[1].splitlines()[0].strip()}")
ref = fx["gamma"]
rows = []
for th in (1, 8, 32):
    torch.set_num_threads(th)
    with torch.no_grad():
        gb = orc.gamma_forward(sd, tau).view(-1).numpy()
        gr = np.array([float(orc.gamma_forward(sd, tau[k:k + 1])[0, 0]) for k in range(0, T + 1, 7)])
    rows.append((th, gb, gr))
    print(f"threads={th:3d}: fp32 batch vs reference-container table: max|dgamma| {np.abs(gb - ref).max():.2e}, "
          f"sigma2_t|s max rel diff {np.max(np.abs(s2(gb) - s2(ref)) / np.abs(s2(ref))):.2%}; "
          f"batch vs row-by-row on this host: max|dgamma| {np.abs(gb[::7] - gr).max():.2e}")
print(f"fp64-evaluated (product) vs reference-container fp32 table: max|dgamma| {np.abs(g64 - ref).max():.2e}, "
      f"sigma2 max rel diff {np.max(np.abs(s2(g64) - s2(ref)) / np.abs(s2(ref))):.2%}")
print(f"fp64-evaluated vs this host's fp32: max|dgamma| {np.abs(g64 - rows[0][1]).max():.2e}, "
      f"sigma2 max rel diff {np.max(np.abs(s2(g64) - s2(rows[0][1])) / np.abs(s2(g64))):.2%}")
