#!/usr/bin/env python3
"""Throughput of the coarse-grained reverse-diffusion sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: draw z_T, 1000 posterior steps (1000 EGNN
dynamics forwards + posterior updates), one more forward + the final decode, results copied to the
host -- i.e. DiffusionQM9.sample() for `--batch` molecules (endiffusion/train_module/diffusion_qm9.py:347-395).
Metric (BASELINE.json): sampled molecules/s, 1000 diffusion steps, B=256 per GPU, N=30 fragments,
production model H=256, L=6 (endiffusion/conf/model/ddpmgblur.yaml).  Synthetic inputs (all-valid padded
point sets, counter-based Gaussian noise) and deterministic random-init weights: the reference ships no
checkpoint and there is no network.

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU: rank 0's weights
are broadcast once over RCCL (xGMI); batches are independent, so there is no per-step collective (weak
scaling, 256 molecules per GPU).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md, dense matrix peaks: v_mfma_f32_32x32x2_f32 157.3 TFLOP/s,
# bf16 MFMA ~2.5 PFLOP/s.  The bf16x3 path executes 3 bf16 MFMA flops per algorithmic flop.
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0}


def edge_flops_per_launch(n_edges: int, H: int) -> float:
    """Algorithmic FLOPs of one edge-kernel launch (one sub-MLP over all valid edges), SURVEY.md
    section 8d: per edge 2H^2 (second Linear) + 2H (attention / coordinate head dot) + 5H (factorised
    first layer: 3 adds + 2 FMAs... counted as in the survey: f_e/(S+1) = 2H^2 + 2H + 5H)."""
    return float(n_edges) * (2.0 * H * H + 2.0 * H + 5.0 * H)


def forward_flops(n_edges: int, n_nodes: int, H: int, L: int, S: int, fin: int) -> float:
    """F_alg of one dynamics forward."""
    f_e = (S + 1) * (2.0 * H * H + 2.0 * H + 5.0 * H)
    f_n = S * 6.0 * H * H + (S + 1) * 4.0 * H * H
    return L * (n_edges * f_e + n_nodes * f_n) + n_nodes * 4.0 * fin * H


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="molecules per GPU per step")
    ap.add_argument("--nodes", type=int, default=30)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--precision", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="matrix-core arithmetic of the H x H contractions (both meet the 1e-4 parity bar)")
    ap.add_argument("--graph", action="store_true", help="replay each diffusion step from a captured hipGraph")
    ap.add_argument("--no-graph", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket the edge-kernel launches of every k-th forward with HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket edge-kernel launches with HIP events in the timed region")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist  # type: ignore
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from hierdiff_amd import DiffusionQM9, _lib, default_config
    from hierdiff_amd.sharding import broadcast_model_weights, shard_sample_ids
    from hierdiff_amd.weights import synthetic_state_dict

    H, L, S, B, N, T = args.hidden, args.layers, 2, args.batch, args.nodes, args.timesteps
    model = DiffusionQM9(default_config(hidden_nf=H, n_layers=L, timesteps=T))
    if rank == 0:
        sd = synthetic_state_dict(9, 0, H, L, S, True, seed=0, coord_gain=1.0)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model = model.to(dev)
    model.dynamics.precision = args.precision
    if world > 1:
        broadcast_model_weights(model, src=0)          # one RCCL broadcast of the packed parameters
    # The timed region brackets the dominant kernel's launches with HIP events (roofline.achieved), which a
    # hipGraph replay cannot carry, so it uses plain launches unless --graph is given; at this size the
    # two are equally fast (6.16 vs 6.18 ms/forward measured) because the GPU, not the host, is the limit.
    model.use_graph = bool(args.graph) and not args.no_graph
    node_mask = torch.ones(B, N, 1, dtype=torch.bool, device=dev)

    lib = _lib.load()
    handle = model._lib_handle()
    topo = model.dynamics.topology(node_mask, None, B, N)
    info = topo.info()

    def one_step(step_idx: int):
        base = shard_sample_ids(step_idx * world * B, world * B, rank, world)[0]
        x, h = model.sample_from_masks(node_mask, None, None, sample_id_base=base)
        return x.cpu(), h.cpu()

    use_events = not args.no_kernel_events
    for w in range(args.warmup):
        one_step(w)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    if use_events:
        _lib.check(lib.hd_profile_enable(handle, 1 | (max(1, args.event_stride) << 8)), "hd_profile_enable")
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    roofline = None
    if use_events:
        ms = (C.c_double * 3)()
        cnt = (C.c_longlong * 3)()
        _lib.check(lib.hd_profile_read(handle, ms, cnt), "hd_profile_read")
        _lib.check(lib.hd_profile_enable(handle, 0), "hd_profile_enable")
        if cnt[0] > 0:
            avg_s = ms[0] / cnt[0] * 1e-3
            fl = edge_flops_per_launch(info["edges"], H)
            achieved = fl / avg_s / 1e12
            traffic = None
            tpath = os.path.join(REPO, "profiles", "edge_kernel_traffic.json")
            if os.path.exists(tpath) and (B, N, H) == (256, 30, 256):
                with open(tpath) as fh:
                    traffic = json.load(fh).get("hbm_bytes_per_launch")
            peak = MFMA_PEAK_TFLOPS[args.precision]
            roofline = {"bound": "mfma", "kernel": "k_edge<256> (GCL + coordinate variants)",
                        "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic,
                        "launches": int(cnt[0]), "avg_launch_us": round(avg_s * 1e6, 2),
                        "flops_per_launch": fl}
            if args.precision == "bf16x3":
                # fp32 operands are split head+tail: each algorithmic flop costs 3 bf16 MFMA flops
                roofline["executed_mfma_tflops"] = round(3 * achieved, 2)
                roofline["executed_frac"] = round(3 * achieved / peak, 4)
                roofline["note"] = ("fp32-accurate contraction emulated with 3 bf16 MFMAs per product (bf16x3); "
                                    "achieved counts algorithmic flops; a pure bf16 MFMA loop on random data "
                                    "sustains 1734 TFLOP/s on this chip (scratch/mb/mb.hip)")

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_fwd = T + 1
    mols = world * B * args.steps
    value = mols / elapsed
    fwd_fl = forward_flops(info["edges"], info["nodes"], H, L, S, 9)
    out = {
        "metric": "sampled molecules/sec (1000 diffusion steps, B=256, N=30)",
        "value": round(value, 3), "unit": "molecules/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16x3" if args.precision == "bf16x3" else "f32", "data": "synthetic",
        "config": {"workload": f"DiffusionQM9.sample: {T}-step reverse diffusion + decode ({n_fwd} EGNN forwards), "
                               f"B={B} per GPU, N={N} all valid, H={H}, L={L}, S=2",
                   "batch_per_gpu": B, "n_nodes": N, "hidden_nf": H, "n_layers": L, "timesteps": T,
                   "precision": args.precision,
                   "precision_note": ("fp32 operands split into bf16 head + tail, 3 bf16 MFMAs per product, fp32 "
                                      "accumulate; <= 1.3e-5 rel-L2 per forward vs the reference (bar 1e-4)")
                   if args.precision == "bf16x3" else "exact fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                   "launch": "hipGraph replay" if model.use_graph else "plain launches",
                   "parallelism": f"{world} independent shards, RCCL weight broadcast only"},
        "ms_per_forward": round(elapsed / args.steps / n_fwd * 1e3, 4),
        "model_tflops": round(fwd_fl * n_fwd * args.steps * world / elapsed / 1e12, 2),
    }
    if roofline:
        out["roofline"] = roofline
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(H, L, N, T)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(H: int, L: int, N: int, T: int) -> dict:
    """The oracle (a structural restatement of the reference's PyTorch-CPU op sequence) timed on this
    host: a bounded sample of EGNN forwards, extrapolated to a full sample (T+1 forwards)."""
    from hierdiff_amd.weights import synthetic_state_dict
    from oracle import egnn_oracle as orc
    threads = os.cpu_count() or 1
    try:
        threads = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:                                   # one torch thread per physical core (SMT siblings do not help)
        import psutil
        phys = psutil.cpu_count(logical=False)
        if phys:
            threads = max(1, min(threads, phys))
    except Exception:
        pass
    Bc = 64
    sd = orc.as_torch_sd(synthetic_state_dict(9, 0, H, L, 2, True, seed=0, coord_gain=1.0))
    cfg = orc.DynCfg(hidden_nf=H, n_layers=L)
    # torch's intra-op pool scales poorly on big dual-socket hosts (128 threads measured 3x SLOWER than 8
    # on this op mix): probe a few pool sizes with one forward each and time the fastest.
    cands = sorted({c for c in (8, 16, 32, threads) if c <= threads})
    probe = {c: orc.time_cpu_forward(sd, cfg, Bc, N, 8, 1, c, time.perf_counter) for c in cands}
    threads = min(probe, key=probe.get)
    reps = 2
    sec = orc.time_cpu_forward(sd, cfg, Bc, N, 8, reps, threads, time.perf_counter)
    return {"value": round(Bc / ((T + 1) * sec), 5), "unit": "molecules/s", "cores": threads, "kind": "port",
            "sample": f"{reps} EGNN dynamics forwards (after 1 warm-up; pool size chosen from {cands} by a 1-forward probe) at B={Bc}, N={N}, H={H}, L={L}, fp32, "
                      f"{threads} torch threads; {sec:.3f} s/forward, extrapolated x{T + 1} forwards per batch",
            "s_per_forward": round(sec, 4)}


if __name__ == "__main__":
    main()
