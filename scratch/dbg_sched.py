import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.helpers import load, rel_l2
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
from hierdiff_amd.noise_model import schedule_tables
from hierdiff_amd import DiffusionQM9, default_config
fx = load("f4_schedule")
sd_np = synthetic_state_dict(9, 0, 32, 1, 2, True, 0)
sd = orc.as_torch_sd(sd_np)
tab = orc.schedule_table(sd, 1000)
print("oracle fp32 here vs fixture: gamma maxabs", np.abs(tab["gamma"]-fx["gamma"]).max())
for k in ("sigma_s","sigma_t","alpha_t_given_s","sigma2_t_given_s"):
    print(k, rel_l2(tab[k], fx[k]), np.abs(tab[k]-fx[k]).max(), "max rel", np.max(np.abs(tab[k]-fx[k])/np.abs(fx[k])))
m = DiffusionQM9(default_config(hidden_nf=32, n_layers=1))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
t64 = schedule_tables(m.gamma, 1000)
print("fp64 vs fixture gamma", np.abs(t64["gamma"].numpy()-fx["gamma"]).max(), "vs oracle-here", np.abs(t64["gamma"].numpy()-tab["gamma"]).max())
print("sigma2 fp64 vs fixture max rel", np.max(np.abs(t64["coef"][:,1].numpy()-fx["sigma2_t_given_s"])/fx["sigma2_t_given_s"]))
print("monotone fixture", np.all(np.diff(fx["gamma"])>0), "here", np.all(np.diff(tab["gamma"])>0), "fp64", np.all(np.diff(t64["gamma"].numpy())>0))
