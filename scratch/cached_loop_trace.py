"""Per-step host times of the plain cached-mask training loop (no staging), 40 steps, one sync at the end."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
exec(open('scratch/fresh_masks_trace.py').read().split("K = 40")[0])
def step_on(bt):
    opt.zero_grad(set_to_none=True)
    loss = m.training_step(bt, 0)
    loss.backward()
    opt.step()
same = {k: v.to(dev) for k, v in ragged(np.arange(B)).items()}
for _ in range(5): step_on(same)
torch.cuda.synchronize()
hs = []; t0 = time.perf_counter()
for k in range(40):
    a = time.perf_counter(); step_on(same); hs.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
print(f"cached loop: {(time.perf_counter() - t0) / 40 * 1e3:.2f} ms/step; host ms per step:", " ".join(f"{x:.0f}" for x in hs))
