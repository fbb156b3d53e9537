"""GPU idle gaps in the staged fresh-mask loop: device events of 8 steps (torch profiler), gaps > 2 ms with their neighbours,
and the host-side calls that overlap each gap."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
exec(open('scratch/fresh_masks_trace.py').read().split("K = 40")[0])
from torch.profiler import profile, ProfilerActivity
batches = [ragged(rng.permutation(B)) for _ in range(30)]
def step_on(bt):
    opt.zero_grad(set_to_none=True)
    loss = m.training_step(bt, 0)
    loss.backward()
    opt.step()
cur = m.stage_batch(batches[0], dev)
for k in range(20):
    step_on(cur); cur = m.stage_batch(batches[k + 1], dev)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for k in range(20, 28):
        step_on(cur); cur = m.stage_batch(batches[k + 1], dev)
    torch.cuda.synchronize()
evs = prof.events()
devs = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
cpus = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
print("device events", len(devs), "cpu events", len(cpus))
end = devs[0].time_range.end
for a, b in zip(devs, devs[1:]):
    gap = b.time_range.start - end
    if gap > 2000:
        over = [c for c in cpus if c.time_range.start < b.time_range.start and c.time_range.end > end and c.time_range.elapsed_us() > 1000]
        over.sort(key=lambda c: -c.time_range.elapsed_us())
        print(f"gap {gap/1e3:.1f} ms after {a.name[:50]} before {b.name[:50]}; long host calls:",
              [(c.name[:40], round(c.time_range.elapsed_us() / 1e3, 1)) for c in over[:6]])
    end = max(end, b.time_range.end)
print("---- neighbourhood of the big gaps")
end = devs[0].time_range.end
t00 = devs[0].time_range.start
for i, (a, b) in enumerate(zip(devs, devs[1:])):
    gap = b.time_range.start - end
    if gap > 20000:
        for e in devs[max(0, i - 6):i + 5]:
            print(f"   dev {(e.time_range.start - t00)/1e3:9.2f} ms  dur {e.time_range.elapsed_us():8.1f} us  {e.name[:70]}")
        lo, hi = end - 3000, b.time_range.start + 500
        near = sorted([c for c in cpus if lo < c.time_range.start < hi and (c.name.startswith('hip') or c.time_range.elapsed_us() > 200)], key=lambda c: c.time_range.start)
        for c in near[:60]:
            print(f"      cpu {(c.time_range.start - t00)/1e3:9.2f} ms  dur {c.time_range.elapsed_us():8.1f} us  {c.name[:60]}")
    end = max(end, b.time_range.end)
print("---- longest device events")
for e in sorted(devs, key=lambda e: -e.time_range.elapsed_us())[:12]:
    print(f"   dev {(e.time_range.start - t00)/1e3:9.2f} ms  dur {e.time_range.elapsed_us():9.1f} us  {e.name[:80]}")
