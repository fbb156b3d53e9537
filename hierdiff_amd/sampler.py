"""Command-line sampler with the reference's wire format.

Replaces `endiffusion/sampler.py:24-41` without Hydra: build the model from the production hyper-parameters
(`conf/model/ddpmgblur.yaml`), load a reference Lightning checkpoint's `state_dict` (keys unchanged, a leading
`model.` prefix is stripped as the reference does, sampler.py:31-32), sample `batch_size x num_batches`
molecules and write `sample_results.pkl` = `pickle((results, test_names))` with
`results: list[{'x': FloatTensor[n,3], 'h': FloatTensor[n,8] (, 'context': FloatTensor[n,1])}]` — exactly what
`generation/ar_sampling_nosize.py:328-329` (`pickle.load(f)[0]`) consumes.

    python -m hierdiff_amd.sampler --checkpoint diffusion.ckpt --batch-size 256 --num-batches 4 --out sample_results.pkl

Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N`; rank 0's weights are broadcast once,
each rank samples a contiguous share of the global sample ids and writes `<out>.rank<r>`; rank 0 concatenates
them in id order into `<out>`.
"""
from __future__ import annotations

import argparse
import os
import pickle
from typing import Dict, List, Optional, Tuple

import torch

from .diffusion import DiffusionQM9, default_config
from .sharding import broadcast_model_weights, shard_sample_ids


def load_reference_state_dict(path: str, trust_checkpoint: bool = False) -> Dict[str, torch.Tensor]:
    """`torch.load(ckpt)['state_dict']` with the leading `model.` prefix removed (sampler.py:27-32).  Tensors that
    do not belong to the sampling half (optimizer state is not in `state_dict`; `pocket_embed.*` only exists for
    pocket models) are passed through untouched and rejected by load_state_dict if unexpected.

    The file is read with `weights_only=True` (tensors and plain containers only).  A Lightning checkpoint that
    carries pickled hyper-parameter objects needs the unrestricted unpickler, which executes code from the file:
    that path is taken only with `trust_checkpoint=True` (CLI: --trust-checkpoint)."""
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as exc:
        if not trust_checkpoint:
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(exc).__name__}: {exc}); pass "
                               "--trust-checkpoint to unpickle it without restrictions (runs code from the file)") from exc
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}


def _attr(node):
    """yaml mapping -> AttrDict, recursively (what DiffusionQM9 reads its OmegaConf node through)."""
    from .diffusion import AttrDict
    if isinstance(node, dict):
        return AttrDict({k: _attr(v) for k, v in node.items()})
    if isinstance(node, list):
        return [_attr(v) for v in node]
    if isinstance(node, str):
        # PyYAML (YAML 1.1) reads `1e-4` as a string, OmegaConf - what the reference loads the file with - as a float
        import re
        if re.fullmatch(r"[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+", node):
            return float(node)
    return node


def load_model_config(path: str):
    """The reference's own model YAML (`endiffusion/conf/model/ddpmgblur.yaml`: `_target_` + a `cfg:` block, which Hydra hands
    to `DiffusionQM9.__init__` as `cfg`, sampler.py:20-22) -> the AttrDict this package's DiffusionQM9 takes.  `analyze` (the
    node-count histogram, `conf/analyze/GEOM.yaml`) is resolved the way a Hydra run does - relative to the directory that holds
    `conf/` - and falls back to the built-in copy of that histogram when the file is not there."""
    import yaml
    with open(path) as fh:
        doc = yaml.safe_load(fh)
    if not isinstance(doc, dict):
        raise ValueError(f"{path}: not a mapping")
    target = doc.get("_target_")
    if target is not None and not str(target).endswith("DiffusionQM9"):
        raise ValueError(f"{path}: _target_ {target!r} is not the coarse-grained diffusion model")
    cfg = _attr(doc.get("cfg", doc))
    for key in ("dynamics", "timesteps", "noise_schedule", "node_coarse_type"):
        if key not in cfg:
            raise ValueError(f"{path}: missing key {key!r} (expected the layout of conf/model/ddpmgblur.yaml)")
    an = cfg.get("analyze")
    if isinstance(an, str) and not os.path.isabs(an):
        here = os.path.dirname(os.path.abspath(path))
        tried = [os.path.join(base, an) for base in (os.path.dirname(os.path.dirname(here)), os.path.dirname(here), here, os.getcwd())]
        found = [p for p in tried if os.path.isfile(p)]
        cfg["analyze"] = found[0] if found else None
    return cfg


def load_sample_config(path: str) -> Tuple[int, int]:
    """`conf/sample/default.yaml` -> (batch_size, num_batches), the keyword arguments of `sample_batches` (sampler.py:38)."""
    import yaml
    with open(path) as fh:
        doc = yaml.safe_load(fh) or {}
    return int(doc["batch_size"]), int(doc["num_batches"])


def write_results(path: str, results: List[dict], test_names: Optional[list] = None) -> None:
    """The reference's output file: one pickle holding the tuple (results, test_names) (sampler.py:39-41)."""
    with open(path, "wb") as f:
        pickle.dump((results, [] if test_names is None else test_names), f)


def read_results(path: str) -> Tuple[List[dict], list]:
    with open(path, "rb") as f:
        res = pickle.load(f)
    return res[0], res[1]


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--checkpoint", default=None, help="reference Lightning checkpoint (.ckpt); random init if omitted")
    ap.add_argument("--batch-size", type=int, default=2)        # conf/sample/default.yaml:1
    ap.add_argument("--num-batches", type=int, default=16)      # conf/sample/default.yaml:2
    ap.add_argument("--out", default="sample_results.pkl")
    ap.add_argument("--hidden-nf", type=int, default=256)
    ap.add_argument("--n-layers", type=int, default=6)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--context", type=float, nargs="*", default=None,
                    help="context values cycled over batches (needs a model with context_node_nf=1)")
    ap.add_argument("--precision", choices=["fp32", "fp16x3"], default="fp32",
                    help="fp32: exact fp32 matrix instructions (the reference's arithmetic); fp16x3: fp32-accurate two-way FP16 "
                         "split on the matrix cores, ~2.4x faster (recommended for sampling)")
    ap.add_argument("--trust-checkpoint", action="store_true",
                    help="allow the unrestricted unpickler for checkpoints that weights_only=True rejects")
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--model-config", default=None,
                    help="the reference's model YAML (conf/model/ddpmgblur.yaml); overrides --hidden-nf / --n-layers / --timesteps")
    ap.add_argument("--sample-config", default=None,
                    help="the reference's sample YAML (conf/sample/default.yaml: batch_size, num_batches)")
    args = ap.parse_args(argv)
    if args.sample_config:
        args.batch_size, args.num_batches = load_sample_config(args.sample_config)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ctx_nf = 1 if args.context else 0
    if args.model_config:
        cfg = load_model_config(args.model_config)
        if ctx_nf:
            cfg.dynamics.context_node_nf = ctx_nf
        model = DiffusionQM9(cfg)
    else:
        model = DiffusionQM9(default_config(hidden_nf=args.hidden_nf, n_layers=args.n_layers, context_node_nf=ctx_nf,
                                            timesteps=args.timesteps))
    if rank == 0 and args.checkpoint:
        model.load_state_dict(load_reference_state_dict(args.checkpoint, args.trust_checkpoint))
    model = model.to(dev)
    model.dynamics.precision = args.precision
    model.seed = args.seed
    if world > 1:
        broadcast_model_weights(model, src=0)

    torch.manual_seed(args.seed)           # the node-count draw uses torch's CPU generator (distributions.py)
    results: List[dict] = []
    first, count = shard_sample_ids(0, args.num_batches, rank, world)      # whole batches per rank
    for b in range(args.num_batches):
        if not (first <= b < first + count):
            # advance the node-count generator over batches owned by other ranks, so the global sequence of
            # molecule sizes (and, with the counter-based noise, every sample) is independent of the world size
            model.nodes_dist.sample(args.batch_size)
            continue
        ctx = None if not args.context else args.context[b % len(args.context)]
        results.extend(model.sample(args.batch_size, dev, context=ctx, sample_id_base=b * args.batch_size))
    if world == 1:
        write_results(args.out, results)
        return 0
    write_results(f"{args.out}.rank{rank}", results)
    import torch.distributed as dist
    dist.barrier()
    if rank == 0:
        merged: List[dict] = []
        for r in range(world):
            merged.extend(read_results(f"{args.out}.rank{r}")[0])
            os.remove(f"{args.out}.rank{r}")
        write_results(args.out, merged)
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
