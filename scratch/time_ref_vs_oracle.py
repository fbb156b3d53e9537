import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import make_golden as mg
from oracle import egnn_oracle as orc
D = mg.import_reference()
model, sd, ocfg = mg.build_reference(D, 256, 6, 0, 0, 1.0)
xh, nm, em = orc.random_inputs([30]*64, 8, 1)
t = torch.full((64,1), 0.5)
torch.set_num_threads(8)
with torch.no_grad():
    for name, fn in (("reference", lambda: model.dynamics._forward(t, xh.clone(), nm, em, None, 30)),
                     ("oracle", lambda: orc.dynamics_forward(sd, ocfg, t, xh, nm, em, None, 30, prefix="dynamics.egnn."))):
        fn(); t0 = time.perf_counter()
        for _ in range(3): fn()
        print(name, (time.perf_counter()-t0)/3)
