"""Multi-GPU sampling: independent sample shards, one process per GPU.

Every molecule of a batch is independent (no cross-sample op anywhere on the path),
so N GPUs run N shards with no data-path collective.  The only collective is one broadcast of rank 0's
parameters at start-up (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests).  Noise comes
from a counter-based generator keyed by the GLOBAL sample id, so a sample does not depend on which rank
(or how many ranks) produced it.
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_sample_ids(first_id: int, total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of the global ids [first_id, first_id + total) over `world` ranks.
    Returns (first id of this rank, count); the first `total % world` ranks get one extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(total, world)
    count = q + (1 if rank < r else 0)
    start = first_id + rank * q + min(rank, r)
    return start, count


def shard_sizes(total: int, world: int) -> List[int]:
    return [shard_sample_ids(0, total, r, world)[1] for r in range(world)]


def pack_parameters(module: torch.nn.Module) -> torch.Tensor:
    """All parameters and buffers of `module` flattened into one fp32 tensor (registration order)."""
    tensors = list(module.parameters()) + list(module.buffers())
    return torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])


@torch.no_grad()
def unpack_parameters(module: torch.nn.Module, flat: torch.Tensor) -> None:
    tensors = list(module.parameters()) + list(module.buffers())
    if sum(t.numel() for t in tensors) != flat.numel():
        raise ValueError("parameter blob size mismatch")
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
        off += n


@torch.no_grad()
def broadcast_model_weights(module: torch.nn.Module, src: int = 0) -> int:
    """One broadcast of the packed parameters from `src` (23.7 MB at L=6); returns the element count."""
    import torch.distributed as dist
    flat = pack_parameters(module).contiguous()
    dist.broadcast(flat, src=src)
    unpack_parameters(module, flat)
    return int(flat.numel())


@torch.no_grad()
def allreduce_gradients(module: torch.nn.Module, average: bool = True) -> int:
    """Data-parallel gradient synchronisation (the reference trains with Lightning DDP, conf/trainer/default.yaml:2-3):
    every gradient is packed into ONE flat fp32 buffer (23.7 MB at L=6), summed over ranks with a single all-reduce
    (RCCL ring over xGMI under backend "nccl": one large message per step is the per-link-friendly shape; "gloo" in the
    CPU tests), divided by the world size and scattered back.  Parameters without a gradient contribute zeros so that
    all ranks reduce the same layout.  Returns the number of elements reduced."""
    import torch.distributed as dist
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return 0
    dev = params[0].device
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return int(flat.numel())


class GradientBuckets:
    """Gradient averaging overlapped with `backward()` (what the reference gets from Lightning's DDP, conf/trainer/default.yaml:2-3).

    The parameters are grouped by EGNN block - `dynamics.egnn.e_block_<i>.*` is one bucket of 987,906 floats (3.95 MB) at H = 256,
    everything else (embedding, output layer, schedule network, pocket embedding) a last one - and each bucket owns ONE flat fp32
    buffer.  Backward produces the blocks' gradients back to front; a post-accumulate hook on every parameter copies its gradient
    into the bucket's buffer, and the bucket's all-reduce is launched (async) the moment its last gradient has landed, so block
    L-1's ring pass runs while blocks L-2 .. 0 are still being differentiated.  `finish()` launches whatever was not complete
    (a parameter without a gradient this step contributes zeros, so every rank reduces the same layout), waits, divides by the
    world size and scatters the averages back into `.grad`.  Several backward passes before one `finish()` (gradient accumulation) are
    handled - a bucket that was sent early and then accumulated into is re-packed and re-sent - at the price of the overlap.

    xGMI note: a ring all-reduce is per-link bound (about 153 GB/s a link); 3.95 MB buckets are ~50 us of wire time each, above
    RCCL's latency floor, and six of them hide behind ~6 ms of backward at the reference's batch size.  Unmeasured on hardware:
    no multi-GPU node was available to this build.  Results are bit-equal to `allreduce_gradients` (the per-element sum over ranks
    does not depend on how elements are grouped into messages): tests/test_sharding_cpu.py.
    """

    def __init__(self, module: torch.nn.Module, average: bool = True):
        import re
        self.average = average
        self.params = [p for _, p in module.named_parameters() if p.requires_grad]
        groups = {}
        for name, p in module.named_parameters():
            if not p.requires_grad:
                continue
            m = re.search(r"e_block_(\d+)\.", name)
            groups.setdefault(int(m.group(1)) if m else -1, []).append(p)
        # launch order = the order backward finishes them: highest block first, the remainder (embedding is differentiated last) last
        self.buckets = [groups[k] for k in sorted(groups, reverse=True) if k >= 0] + ([groups[-1]] if -1 in groups else [])
        self._slot = {}
        for b, plist in enumerate(self.buckets):
            off = 0
            for p in plist:
                self._slot[id(p)] = (b, off)
                off += p.numel()
        self._sizes = [sum(p.numel() for p in plist) for plist in self.buckets]
        self._flat = [None] * len(self.buckets)
        self._ready = [0] * len(self.buckets)
        self._seen = [set() for _ in self.buckets]
        self._work = [None] * len(self.buckets)
        self._dirty = [False] * len(self.buckets)
        self.launch_order = []                   # bucket indices in the order their all-reduce was issued (read by the tests)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _buffer(self, b: int, like: torch.Tensor) -> torch.Tensor:
        f = self._flat[b]
        if f is None or f.device != like.device:
            f = self._flat[b] = torch.zeros(self._sizes[b], dtype=torch.float32, device=like.device)
        return f

    def _on_grad(self, p: torch.Tensor) -> None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        b, off = self._slot[id(p)]
        if id(p) in self._seen[b] or self._work[b] is not None:      # a second accumulation into the same parameter (shared weights,
            self._dirty[b] = True                                     # several backward passes per step): what is in the buffer -
            return                                                    # or already on the wire - is stale; finish() re-packs and re-sends
        self._buffer(b, p.grad)[off:off + p.numel()].copy_(p.grad.reshape(-1))
        self._seen[b].add(id(p))
        if len(self._seen[b]) == len(self.buckets[b]):
            self._launch(b)

    def _launch(self, b: int) -> None:
        import torch.distributed as dist
        self._work[b] = dist.all_reduce(self._flat[b], op=dist.ReduceOp.SUM, async_op=True)
        self.launch_order.append(b)

    @torch.no_grad()
    def finish(self) -> int:
        """Complete the step's reduction; returns the number of elements reduced.  Collective: every rank calls it once per step."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return 0
        world = dist.get_world_size()
        for b, plist in enumerate(self.buckets):
            if self._dirty[b] and self._work[b] is not None:
                self._work[b].wait()                 # the early all-reduce carried a partial sum: let it land, then send the final one
                self._work[b] = None
            if self._work[b] is None:                # incomplete / re-accumulated bucket: pack what exists, zeros for the rest
                flat = self._buffer(b, plist[0])
                off = 0
                for p in plist:
                    n = p.numel()
                    if p.grad is None:
                        flat[off:off + n].zero_()
                    else:
                        flat[off:off + n].copy_(p.grad.reshape(-1))
                    off += n
                self._launch(b)
        total = 0
        for b, plist in enumerate(self.buckets):
            self._work[b].wait()
            flat = self._flat[b]
            if self.average:
                flat /= world
            off = 0
            for p in plist:
                n = p.numel()
                g = flat[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
            total += off
            self._work[b] = None
            self._dirty[b] = False
            self._seen[b].clear()
        return total

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
