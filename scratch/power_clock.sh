#!/bin/bash
# socket power / engine clock while the dynamics forward runs back to back (does the power cap set the clock?): scratch/power_clock.sh <precision>
prec=${1:-fp32}
python - "$prec" <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import build_dynamics, DEV
from oracle import egnn_oracle as orc
from hierdiff_amd.weights import synthetic_state_dict
prec = sys.argv[1]
sd_np = synthetic_state_dict(9, 0, 256, 6, 2, True, 0, 1.0)
xh, nm, em = orc.random_inputs([30] * 256, 8, 1)
xh, nm = xh.to(DEV), nm.to(DEV)
t = torch.full((256, 1), 0.5, device=DEV)
dyn = build_dynamics(sd_np, 256, 6); dyn.precision = prec
topo = dyn.topology(nm, None, 256, 30); dyn.sync_weights()
t0 = time.time(); n = 0
while time.time() - t0 < 25:
    for _ in range(50): dyn.forward_with_topology(topo, t, xh, None, None)
    torch.cuda.synchronize(); n += 50
print(prec, "forwards", n, "ms/forward %.3f" % ((time.time() - t0) / n * 1e3))
PY
pid=$!
sleep 12
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|Max Graphics" | head -6
  echo --
  sleep 2
done
wait $pid
