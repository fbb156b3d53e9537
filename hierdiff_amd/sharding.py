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
