"""What stalls a training loop when a topology is created while the GPU is busy?  Cached-mask loop + after each step's launch:
A nothing; B the host layout alone (hd_topology_layout: CPU work, allocations); C a full Topology (layout + pooled arena +
upload + fill) that nobody uses; D = C with the pool trimmed every step (fresh hipMalloc / hipHostMalloc)."""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, '.')
exec(open('scratch/fresh_masks_trace.py').read().split("K = 40")[0])
from hierdiff_amd import _lib
from hierdiff_amd.dynamics import Topology
lib = _lib.load()
def step_on(bt):
    opt.zero_grad(set_to_none=True)
    loss = m.training_step(bt, 0)
    loss.backward()
    opt.step()
same = {k: v.to(dev) for k, v in ragged(np.arange(B)).items()}
hosts = [ragged(rng.permutation(B)) for _ in range(30)]
masks = [(np.ascontiguousarray(b["atom_mask"].reshape(-1).numpy()).view(np.uint8), np.ascontiguousarray(b["edge_mask"].reshape(-1).numpy()).view(np.uint8)) for b in hosts]
for _ in range(5): step_on(same)
torch.cuda.synchronize()
def run(extra, name):
    keep = []
    for k in range(3): step_on(same); extra(k, keep)
    torch.cuda.synchronize(); t0 = time.perf_counter(); hs = []
    for k in range(3, 28):
        step_on(same)
        a = time.perf_counter(); extra(k, keep); hs.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 25 * 1e3:.2f} ms/step; extra host ms median {np.median(hs):.2f} max {max(hs):.1f}")
def layout(k, keep):
    nm, em = masks[k]
    counts = (C.c_longlong * 5)()
    _lib.check(lib.hd_topology_layout(nm.ctypes.data, em.ctypes.data, B, N, counts, None, None, None, None, None, None), "layout")
def topo(k, keep):
    nm, em = masks[k]
    keep.append(Topology(m.dynamics, None, None, B, N, host_masks=(nm, em)))
    if len(keep) > 8: keep.pop(0)
def topo_trim(k, keep):
    topo(k, keep); keep.clear(); lib.hd_arena_pool_trim()
run(lambda k, keep: None, "A nothing")
run(layout, "B host layout only")
run(topo, "C topology (pooled)")
run(lambda k, keep: None, "A nothing")
run(topo_trim, "D topology, fresh allocations")
def run2(name, use_staged, stage=True):
    cur = m.stage_batch(hosts[0], dev)
    for k in range(3):
        step_on(cur if use_staged else same); cur = m.stage_batch(hosts[k + 1], dev) if stage else cur
    torch.cuda.synchronize(); t0 = time.perf_counter(); hs = []
    for k in range(3, 28):
        step_on(cur if use_staged else same)
        a = time.perf_counter()
        if stage: cur = m.stage_batch(hosts[k + 1], dev)
        hs.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 25 * 1e3:.2f} ms/step; stage host ms median {np.median(hs):.2f} max {max(hs):.1f}")
run2("F stage_batch every step, step on the cached batch", False)
run2("E stage_batch every step, step on the staged batch", True)
devb = [{k: v.to(dev) for k, v in b.items()} for b in hosts]
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(3, 28): step_on(devb[k])
torch.cuda.synchronize()
print(f"G device-only fresh masks: {(time.perf_counter() - t0) / 25 * 1e3:.2f} ms/step")
run2("E again", True)
from hierdiff_amd.dynamics import _to_device_async
def only_copies(k, keep):
    keep[:] = [_to_device_async(v, dev) for v in hosts[k].values()]
def only_masks(k, keep):
    keep[:] = list(m.dynamics.stage_masks(hosts[k]["atom_mask"], hosts[k]["edge_mask"], dev))
def only_pos(k, keep):
    keep[:] = [_to_device_async(hosts[k]["positions"], dev)]
run(only_copies, "F1 four tensor uploads per step")
run(only_pos, "F1b one tensor upload per step")
run(only_masks, "F2 stage_masks per step")
run(lambda k, keep: None, "A nothing")
for name in ("atom_mask", "edge_mask", "node_feature"):
    run(lambda k, keep, name=name: keep.__setitem__(slice(None), [_to_device_async(hosts[k][name], dev)]), f"H one upload per step: {name}")
def two_pos(k, keep):
    keep[:] = [_to_device_async(hosts[k]["positions"], dev), _to_device_async(hosts[k]["node_feature"], dev)]
run(two_pos, "H2 positions + node_feature")
