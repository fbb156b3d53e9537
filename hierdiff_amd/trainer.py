"""The data-parallel optimisation step around `DiffusionQM9.training_step`, without Lightning.

What the reference's trainer does per batch (`endiffusion/conf/trainer/default.yaml`: strategy ddp, gradient_clip_val 2,
gradient_clip_algorithm norm, accumulate_grad_batches 1; `conf/optim/adamw.yaml`: AdamW lr 4e-4, weight_decay 4e-8;
`conf/scheduler/step.yaml`: StepLR step_size 15, gamma 0.1, stepped once per epoch; `train_module/diffusion_qm9.py:774-777`):

    loss = model.training_step(batch)      # forward + loss on this rank's shard of the batch
    loss.backward()
    <DDP: gradients averaged over ranks>   # here: one all-reduce per EGNN block, launched from backward hooks as the block's
                                           # gradients land (hierdiff_amd.sharding.GradientBuckets, RCCL over xGMI)
    clip_grad_norm_(parameters, 2.0)
    optimizer.step()

One process per GPU (`python -m torch.distributed.run --nproc-per-node N ...`); at world size 1, or without an initialised
process group, the all-reduce is skipped.  This module holds no training loop policy beyond that step: data loading, logging and
checkpointing stay with the caller.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

from .sharding import GradientBuckets, allreduce_gradients


def configure_optimizers(model: torch.nn.Module, lr: float = 4.0e-4, weight_decay: float = 4.0e-8, step_size: int = 15,
                         gamma: float = 0.1, fused: Optional[bool] = None):
    """(AdamW, StepLR) with the reference's values (conf/optim/adamw.yaml, conf/scheduler/step.yaml).  On the GPU the update runs as
    torch's fused multi-tensor AdamW (`fused=True`: three launches for the model's 60-odd parameter tensors instead of ~15 foreach
    launches; 1.5 -> 0.5 ms of host time per step, which is what a step at the reference's batch size of 16 is made of -
    scratch/train_host_time.py); same update rule, `fused=False` for the foreach form."""
    params = list(model.parameters())
    if fused is None:
        fused = bool(params) and all(p.is_cuda for p in params)
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, **({"fused": True} if fused else {}))
    return opt, torch.optim.lr_scheduler.StepLR(opt, step_size=step_size, gamma=gamma)


def _world() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _launch_step(model, batch: Dict[str, torch.Tensor], optimizer: torch.optim.Optimizer, clip_val: Optional[float],
                 overlap: bool = True):
    """Everything of one optimisation step that is queued on the GPU; returns the (loss, grad norm) device scalars."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    buckets = None
    if _world() > 1 and overlap:
        buckets = getattr(model, "_grad_buckets", None)
        if buckets is None:                      # hooks registered once per model; they stay silent at world size 1
            buckets = GradientBuckets(model, average=True)
            model._grad_buckets = buckets
    loss = model.training_step(batch, 0)
    loss.backward()
    if buckets is not None:
        buckets.finish()
    elif _world() > 1:
        allreduce_gradients(model, average=True)
    params = [p for p in model.parameters() if p.grad is not None]
    if clip_val is not None:
        norm = torch.nn.utils.clip_grad_norm_(params, float(clip_val))
    else:
        norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad) for p in params]))
    optimizer.step()
    return loss.detach(), norm


def ddp_step(model, batch: Dict[str, torch.Tensor], optimizer: torch.optim.Optimizer, clip_val: Optional[float] = 2.0,
             overlap: bool = True) -> Dict[str, float]:
    """One optimisation step on this rank's batch; returns {"loss", "grad_norm"} (the norm BEFORE clipping, like Lightning's
    track_grad_norm: 2).  Every rank must call it the same number of times (the all-reduce is collective).  `overlap=False`
    averages with one flat all-reduce after backward instead of the per-block buckets (same bits)."""
    loss, norm = _launch_step(model, batch, optimizer, clip_val, overlap)
    return {"loss": float(loss), "grad_norm": float(norm)}


def fit_epoch(model, batches: Iterable[Dict[str, torch.Tensor]], optimizer, scheduler=None, clip_val: Optional[float] = 2.0,
              device: Optional[torch.device] = None):
    """`ddp_step` over an iterable of reference-style batch dicts (keys positions, atom_mask, edge_mask, node_feature ...), the
    scheduler stepped once at the end (StepLR counts epochs).  Returns the list of per-step dicts.

    With `device` the batches are HOST batches (a DataLoader's) and the loop is pipelined one batch deep: after step k is
    queued, batch k+1 is staged (`model.stage_batch`: pinned copies in stream order and the masks' topology laid out from the
    host copies) BEFORE the host waits for step k's loss - new masks every step cost no GPU idle time."""
    log = []
    if device is None:
        for batch in batches:
            log.append(ddp_step(model, batch, optimizer, clip_val))
    else:
        stage = getattr(model, "stage_batch", None)
        if stage is None:
            stage = lambda b, d: {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in b.items()}
        it = iter(batches)
        nxt = next(it, None)
        cur = None if nxt is None else stage(nxt, device)
        while cur is not None:
            loss, norm = _launch_step(model, cur, optimizer, clip_val)
            nxt = next(it, None)
            cur = None if nxt is None else stage(nxt, device)
            log.append({"loss": float(loss), "grad_norm": float(norm)})
    if scheduler is not None:
        scheduler.step()
    return log
