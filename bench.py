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

The JSON line's headline (`value`, `dtype`, `roofline`) is the EXACT-fp32 path (v_mfma_f32_32x32x2_f32), the
arithmetic the reference computes in.  The one opt-in mode is timed in the same invocation over the same K steps and
reported as a sibling block with its own roofline and its measured deviation from the fp32 path: `"fp16x3"`
(per-edge and node contractions on a two-way FP16 split of operands ranged by exact powers of two, 3 fp16 MFMAs per product;
fp32-ACCURATE: as far from a float64 evaluation as the exact-fp32 path and the float32 reference themselves,
tests/test_gpu_parity.py).  (The bf16 splits of rounds 1-5 were retired in round 6.)  At N=1 the line also carries
`cpu_baseline` (the oracle timed on this host) and `configs` (BASELINE.json configs 2, 3, 5 and the reference's
default B=2 job, timed on short chains).

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU: rank 0's weights
are broadcast once over RCCL (xGMI); batches are independent, so there is no per-step collective (weak
scaling, 256 molecules per GPU).  The line reports every rank's own elapsed time (`rank_elapsed_s`) next to the max
over ranks of the barrier-to-barrier time (`elapsed_max_s`, what `value` is computed from), and `rccl_ranks`.  Under
the launcher the process group is initialised at world size 1 too, so the same code path runs in the 1-GPU test tier.
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

# dmabuf IPC (RCCL between the per-GPU ranks fails with `hipIpcGetMemHandle: invalid argument` without it on this driver): set
# in-process before torch / HIP load, so that a rank started by the driver's own launcher line - not only by self_launch() below -
# has it.  hierdiff_amd/__init__.py does the same for every other entry point.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md, dense matrix peaks: v_mfma_f32_32x32x2_f32 157.3 TFLOP/s,
# fp16 MFMA ~2.5 PFLOP/s; HBM3E 8 TB/s.  The fp16x3 path executes 3 fp16 MFMA flops per algorithmic flop.
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "fp16x3": 2500.0}
MFMAS_PER_PRODUCT = {"fp32": 1, "fp16x3": 3}
HBM_PEAK_GBPS = 8000.0
DTYPE = {"fp32": "f32", "fp16x3": "fp16x3"}
# PMC figures are NOT measured by this process (counter passes need rocprofv3 around the run): they are replayed from the
# newest committed summary of scratch/round_profiles.sh + summarize_profiles.py, and only when that summary was collected
# on the very library that is loaded now (sha256 of libhierdiff_hip.so) and on this workload's shape.
COUNTER_FILES = [os.path.join(REPO, "profiles", f) for f in ("r06_counters.json", "r05_counters.json", "r04_counters.json", "r03_counters.json",
                                                              os.path.join("history", "r02_counters.json"))]


def lib_sha256() -> str:
    import hashlib
    from hierdiff_amd import _lib
    with open(_lib.LIB_PATH, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def kernel_source_sha256() -> str:
    """sha256 over the kernel sources the library is built from (csrc/*, include/*.h, the build flags): identifies the
    kernels independently of where and when hipcc produced the binary."""
    import hashlib
    from hierdiff_amd import build as _b
    h = hashlib.sha256()
    for path in sorted(_b.DEPS):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def edge_flops_per_launch(n_edges: int, H: int) -> float:
    """Algorithmic FLOPs of one edge-kernel launch (one sub-MLP over all valid edges), SURVEY.md
    section 8d: per edge 2H^2 (second Linear) + 2H (attention / coordinate head dot) + 5H (factorised
    first layer), f_e/(S+1) = 2H^2 + 2H + 5H."""
    return float(n_edges) * (2.0 * H * H + 2.0 * H + 5.0 * H)


def forward_flops(n_edges: int, n_nodes: int, H: int, L: int, S: int, fin: int) -> float:
    """F_alg of one dynamics forward."""
    f_e = (S + 1) * (2.0 * H * H + 2.0 * H + 5.0 * H)
    f_n = S * 6.0 * H * H + (S + 1) * 4.0 * H * H
    return L * (n_edges * f_e + n_nodes * f_n) + n_nodes * 4.0 * fin * H


def load_counters(precision: str, shape) -> dict:
    """PMC figures of the edge kernel replayed from a committed rocprofv3 summary (HBM bytes per launch, MFMA-busy and
    wait fractions).  Refused - {"stale": reason} - unless the summary names this workload's shape AND the sha256 of the
    library loaded now: a kernel change must not keep old counters alive."""
    reasons = []
    for path in COUNTER_FILES:
        rel = os.path.relpath(path, REPO)
        try:
            with open(path) as fh:
                c = json.load(fh)
        except Exception:
            continue
        if tuple(c.get("shape", [])) != tuple(shape):
            reasons.append(f"{rel}: collected on shape {c.get('shape')}")
            continue
        # the summary names the binary it was collected on AND the kernel sources that binary was built from; a rebuild of
        # the same sources (the driver builds in its own container) is the same kernels
        if c.get("lib_sha256") != lib_sha256() and c.get("kernel_source_sha256") != kernel_source_sha256():
            reasons.append(f"{rel}: collected on library {str(c.get('lib_sha256'))[:12]} / kernel sources "
                           f"{str(c.get('kernel_source_sha256'))[:12]}, loaded {lib_sha256()[:12]} / {kernel_source_sha256()[:12]}")
            continue
        out = dict(c.get("edge_kernel", {}).get(precision, {}))
        if not out:
            continue
        out["replayed_from"] = rel
        out["lib_sha256"] = c.get("lib_sha256", "")
        out["kernel_source_sha256"] = c.get("kernel_source_sha256", "")
        return out
    return {"stale": "; ".join(reasons) or "no counter summary under profiles/"}


def build_model(H, L, T, dev, rank, world, context_nf=0, cls=None, seed=0, dist_on=False):
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.sharding import broadcast_model_weights
    from hierdiff_amd.weights import synthetic_state_dict
    cls = cls or DiffusionQM9
    model = cls(default_config(hidden_nf=H, n_layers=L, context_node_nf=context_nf, timesteps=T))
    if rank == 0:
        sd = synthetic_state_dict(9, context_nf, H, L, 2, True, seed=seed, coord_gain=1.0)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model = model.to(dev)
    if dist_on:
        # one RCCL broadcast of the packed parameters (xGMI between GPUs); also issued at world size 1 when the file
        # runs under the launcher, so the N > 1 code path is exercised by the single-GPU test tier
        model.rccl_broadcast_elements = broadcast_model_weights(model, src=0)
    return model


def timed_headline(model, precision, args, dev, rank, world, dist):
    """W untimed + exactly K timed steps of the headline workload in one precision; returns the result block."""
    from hierdiff_amd import _lib
    from hierdiff_amd.sharding import shard_sample_ids
    H, L, S, B, N, T = args.hidden, args.layers, 2, args.batch, args.nodes, args.timesteps
    model.dynamics.precision = precision
    # The timed region brackets the dominant kernel's launches with HIP events (roofline.achieved), which a
    # hipGraph replay cannot carry, so it uses plain launches unless --graph is given; at this size the two are
    # equally fast because the GPU, not the host, is the limit.
    model.use_graph = bool(args.graph)
    node_mask = torch.ones(B, N, 1, dtype=torch.bool, device=dev)
    lib = _lib.load()
    handle = model._lib_handle()
    topo = model.dynamics.topology(node_mask, None, B, N)
    info = topo.info()

    def one_step(step_idx: int):
        base = shard_sample_ids(step_idx * world * B, world * B, rank, world)[0]
        x, h = model.sample_from_masks(node_mask, None, None, sample_id_base=base)
        return x.cpu(), h.cpu()

    use_events = not args.no_kernel_events and not model.use_graph
    for w in range(args.warmup):
        one_step(w)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    if use_events:
        _lib.check(lib.hd_profile_enable(handle, 3 | (max(1, args.event_stride) << 8)), "hd_profile_enable")
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k)
    torch.cuda.synchronize(dev)
    own = time.perf_counter() - t0            # this rank's K steps, before it waits for the others
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_elapsed = [own]
    if dist is not None:
        # every rank's own time (a straggler shows here) and the max over ranks of the barrier-to-barrier time
        mine = torch.tensor([own, elapsed], device=dev, dtype=torch.float64)
        allr = torch.empty(world * 2, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 2).cpu()
        rank_elapsed = [float(v) for v in allr[:, 0]]
        elapsed = float(allr[:, 1].max())

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
            peak = MFMA_PEAK_TFLOPS[precision]
            pmc = load_counters(precision, (B, N, H, L)) if (B, N, H, L) == (256, 30, 256, 6) else {"stale": "not the headline shape"}
            traffic = pmc.get("hbm_bytes_per_launch")
            roofline = {"bound": "mfma",
                        "kernel": f"k_edge<{H}, *, {precision}> (GCL + coordinate variants)",
                        "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": traffic,
                        "launches": int(cnt[0]), "avg_launch_us": round(avg_s * 1e6, 2),
                        "flops_per_launch": fl, "edge_rows_per_launch": info["tiles"] * 32}
            if traffic:
                # replayed bytes (committed PMC pass on this very library) over this kernel's live average duration
                replay = {"replayed_from": pmc["replayed_from"], "lib_sha256": pmc["lib_sha256"][:16],
                          "kernel_source_sha256": pmc["kernel_source_sha256"][:16],
                          "hbm_bytes_per_launch": traffic, "hbm_gbps": round(traffic / avg_s / 1e9, 1),
                          "hbm_frac": round(traffic / avg_s / 1e9 / HBM_PEAK_GBPS, 4)}
                for k in ("mfma_busy", "valu_issue_frac", "wait_inst_frac", "wait_any_frac"):
                    if k in pmc:
                        replay[k] = pmc[k]
                roofline["pmc"] = replay
            else:
                roofline["pmc"] = {"replayed_from": None, "refused": pmc.get("stale", "no entry for this precision")}
            if precision != "fp32":
                # fp32 operands are split into two fp16 pieces: each algorithmic flop costs 3 fp16 MFMA flops
                m = MFMAS_PER_PRODUCT[precision]
                roofline["executed_mfma_tflops"] = round(m * achieved, 2)
                roofline["executed_frac"] = round(m * achieved / peak, 4)
                roofline["vs_fp32_mfma_peak"] = round(achieved / MFMA_PEAK_TFLOPS["fp32"], 4)
                roofline["note"] = (f"contraction on {m} fp16 MFMAs per product ({precision}); achieved counts algorithmic "
                                    "flops, executed_* the issued MFMA flops; vs_fp32_mfma_peak = achieved / 157.3")
        if cnt[1] > 0:
            # the node side (family 1: the fused k_node_f32 / k_node launches, or k_gemm_r16 below HD_FUSE_MIN_ROWS): algorithmic
            # flops of all node-level Linears of a forward (f_n = S 6H^2 + (S+1) 4H^2 per node and block)
            # over the launches that carry them
            node_s = ms[1] / cnt[1] * 1e-3
            launches_per_fwd = cnt[1] / max(1, cnt[0]) * (L * (S + 1))
            node_fl = L * info["nodes"] * (S * 6.0 * H * H + (S + 1) * 4.0 * H * H) / max(1.0, launches_per_fwd)
            peak = MFMA_PEAK_TFLOPS[precision]
            m = MFMAS_PER_PRODUCT[precision]
            if roofline is not None:
                roofline["node_kernel"] = {"kernel": "k_node_f32" if precision == "fp32" else f"k_node ({precision})",
                                           "achieved": round(node_fl / node_s / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                                           "frac": round(node_fl / node_s / 1e12 / peak, 4),
                                           "executed_frac": round(m * node_fl / node_s / 1e12 / peak, 4),
                                           "avg_launch_us": round(node_s * 1e6, 2), "launches": int(cnt[1]),
                                           "launches_per_forward": round(launches_per_fwd, 2),
                                           "flops_per_launch_avg": node_fl,
                                           "share_of_forward": round(ms[1] / max(1e-9, ms[0] + ms[1] + ms[2]), 4)}
    n_fwd = T + 1
    mols = world * B * args.steps
    fwd_fl = forward_flops(info["edges"], info["nodes"], H, L, S, 9)
    model_tflops = fwd_fl * n_fwd * args.steps * world / elapsed / 1e12
    block = {"value": round(mols / elapsed, 3), "unit": "molecules/s", "dtype": DTYPE[precision], "steps": args.steps,
             "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
             "ms_per_forward": round(elapsed / args.steps / n_fwd * 1e3, 4),
             "model_tflops": round(model_tflops, 2),
             # whole forward (edge + node kernels + everything around them), algorithmic flops over the wall clock,
             # against the dense MFMA peak of the mode, per GPU
             "frac_end_to_end": round(model_tflops / world / MFMA_PEAK_TFLOPS[precision], 4),
             "elapsed_max_s": round(elapsed, 4), "rank_elapsed_s": [round(v, 4) for v in rank_elapsed],
             "launch": "hipGraph replay" if model.use_graph else "plain launches"}
    if roofline:
        sus = sustained_mfma(precision, dev)
        if sus:
            # the chip's own rate for this mode's matrix instruction under its power budget, measured right here (hd_mfma_probe:
            # register operands, random data, two wavefronts per SIMD): what `frac` (against the guide's dense peak at 2.4 GHz)
            # cannot be, and what the kernel's matrix work is priced against in DESIGN.md section 12b
            m = MFMAS_PER_PRODUCT[precision]
            sus["executed_frac_of_sustained"] = round(m * roofline["achieved"] / sus["tflops"], 4)
            sus["matrix_time_us_per_launch"] = round(m * roofline["flops_per_launch"] / (sus["tflops"] * 1e12) * 1e6, 1)
            roofline["sustained"] = sus
        block["roofline"] = roofline
    return block


def sustained_mfma(precision: str, dev) -> dict:
    """TFLOP/s the chip sustains when every SIMD streams the mode's MFMA opcode from registers (include/hierdiff_hip.h:
    hd_mfma_probe); {} if the probe is unavailable.  ~30 ms, outside every timed region."""
    from hierdiff_amd import _lib
    try:
        lib = _lib.load()
        kind = {"fp32": 0, "fp16x3": 1}[precision]
        g = torch.Generator().manual_seed(11)
        data = (torch.rand(1024, generator=g) * 2 - 1).to(dev)
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        scratch = torch.empty(2 * 256 * n_cu, device=dev, dtype=torch.float32)
        ns = C.c_double()
        iters = 3000 if kind == 0 else 12000
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(lib.hd_mfma_probe(idx, kind, data.data_ptr(), scratch.data_ptr(), iters, C.byref(ns),
                                     torch.cuda.current_stream(dev).cuda_stream), "hd_mfma_probe")
        flop_per_mfma = 2.0 * 32 * 32 * (2 if kind == 0 else 16)
        tf = flop_per_mfma / (ns.value * 1e-9) * 4 * n_cu / 1e12
        return {"instruction": ["v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_f16"][kind],
                "ns_per_mfma_per_simd": round(ns.value, 2), "tflops": round(tf, 1),
                "what": "register-operand MFMA loop on every SIMD, 2 wavefronts per SIMD, random operands, measured in this process"}
    except Exception:               # a measurement aid must never fail the bench
        return {}


@torch.no_grad()
def precision_gap(model, args, dev, mode="fp16x3") -> dict:
    """rel-L2 between a split mode and the exact-fp32 mode on one headline-shaped forward (the fp32 path is the yardstick
    here; every mode is checked against the reference-generated golden vectors in tests/)."""
    B, N = args.batch, args.nodes
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, N, 3, generator=g)
    xh = torch.cat([x - x.mean(1, keepdim=True), torch.randn(B, N, 8, generator=g)], dim=2).to(dev)
    nm = torch.ones(B, N, 1, dtype=torch.bool, device=dev)
    worst = 0.0
    outs = {}
    for tv in (0.05, 0.5, 0.95):
        t = torch.full((B, 1), tv, device=dev)
        for p in ("fp32", mode):
            model.dynamics.precision = p
            outs[p] = model.dynamics._forward(t, xh, nm, None, None, None).double()
        worst = max(worst, float(torch.linalg.norm(outs[mode] - outs["fp32"]) / torch.linalg.norm(outs["fp32"])))
    bound = {"fp16x3": "<= 8e-7 rel-L2 per forward on every golden fixture, the exact-fp32 mode's own figure; distance to a float64 "
                       "evaluation 3.5e-7 vs 3.8e-7 (exact fp32) and 3.0e-7 (float32 reference), "
                       "tests/test_gpu_parity.py; operands ranged per matrix / per edge row by exact powers of two: no range assumption"}
    return {"max_rel_l2_vs_fp32_path": float(f"{worst:.3e}"), "bound_vs_reference": bound[mode]}


def other_configs(args, dev) -> dict:
    """BASELINE.json configs 2, 3, 5 and the B=64 / B=2 jobs timed at the FULL chain length (T = 1000 posterior steps +
    decode = 1001 forwards, one untimed pass first, results copied to the host like the headline); the L=9 variant of the
    headline and the pocket-sized graph on a short chain (T_short steps, per-forward cost scaled to
    1001 forwards - marked `extrapolated`)."""
    from hierdiff_amd import EnVariationalDiffusion
    from hierdiff_amd.geom_stats import GEOM_FRAGMENT_HISTOGRAM as HIST
    Ts = args.config_timesteps
    out = {"timesteps_short": Ts, "note": "entries without `extrapolated` are real 1000-step runs: molecules_per_s = B / wall"}

    def timeit(fn, reps=1):
        """One untimed pass, then the fastest of `reps` timed ones (the short-chain entries are ~50 ms each: a deferred
        deallocation of the previous entry's model landing inside one of them would otherwise double it)."""
        import gc
        fn()
        gc.collect()
        torch.cuda.synchronize(dev)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def host(xh):
        return xh[0].cpu(), xh[1].cpu()

    def full(B, dt, L, n_edges=None, n_nodes=None, prec="fp32"):
        e = {"s_per_batch": round(dt, 4), "ms_per_forward": round(dt / 1001 * 1e3, 4), "molecules_per_s": round(B / dt, 2),
             "timesteps": 1000}
        if n_edges is not None:
            tf = forward_flops(n_edges, n_nodes, 256, L, 2, 9) * 1001 / dt / 1e12
            e["model_tflops"] = round(tf, 2)
            e["frac_end_to_end"] = round(tf / MFMA_PEAK_TFLOPS[prec], 4)
        return e

    def short(B, dt, T):
        ms = dt / (T + 1) * 1e3
        return {"ms_per_forward": round(ms, 4), "molecules_per_s": round(B / (ms * 1e-3 * 1001), 2), "extrapolated": True,
                "timesteps": T}

    rng = np.random.Generator(np.random.PCG64(2022))
    keys = np.array([k for k in HIST if k <= 48])
    p = np.array([HIST[k] for k in keys], float)
    n3 = rng.choice(keys, size=256, p=p / p.sum())
    nm3 = (torch.arange(48)[None, :] < torch.tensor(n3)[:, None]).unsqueeze(-1).to(dev)
    e3, v3 = int((n3 * (n3 - 1)).sum()), int(n3.sum())
    nm5 = torch.ones(64, 30, 1, dtype=torch.bool, device=dev)
    em5 = torch.zeros(64, 30, 30, dtype=torch.bool)
    em5[:, :24, :24] = True
    em5[:, 24:, 24:] = True
    em5 = (em5 & ~torch.eye(30, dtype=torch.bool)[None]).to(dev)
    ctx5 = torch.full((64, 30, 1), 2.3, device=dev)
    nm64 = torch.ones(64, 30, 1, dtype=torch.bool, device=dev)
    nm256 = torch.ones(256, 30, 1, dtype=torch.bool, device=dev)
    nm2 = torch.ones(2, 30, 1, dtype=torch.bool, device=dev)
    nmp = torch.ones(32, 200, 1, dtype=torch.bool, device=dev)
    m9s = build_model(256, 9, Ts, dev, 0, 1)
    m6s = build_model(256, 6, Ts, dev, 0, 1)
    m9 = build_model(256, 9, 1000, dev, 0, 1)
    m6 = build_model(256, 6, 1000, dev, 0, 1)
    m5 = build_model(256, 6, 1000, dev, 0, 1, context_nf=1, cls=EnVariationalDiffusion)
    for prec in ("fp32", "fp16x3"):
        for m in (m9s, m6s, m9, m6, m5):
            m.dynamics.precision = prec
        blk = {}
        blk["config2_B64_N30_L9"] = full(64, timeit(lambda: host(m9.sample_from_masks(nm64, None, None))), 9, 64 * 870, 64 * 30, prec)
        blk["config3_B256_geom_sizes_pad48_L6"] = dict(full(256, timeit(lambda: host(m6.sample_from_masks(nm3, None, None))), 6, e3, v3, prec),
                                                       mean_n=round(float(n3.mean()), 2))
        blk["config5_B64_N30_context_fixnoise_mol24_L6"] = full(
            64, timeit(lambda: host(m5.sample(64, 30, nm5, em5, ctx5, fix_noise=True))), 6, 64 * (24 * 23 + 6 * 5), 64 * 30, prec)
        blk["B64_N30_L6"] = full(64, timeit(lambda: host(m6.sample_from_masks(nm64, None, None))), 6, 64 * 870, 64 * 30, prec)
        # the reference's shipped job: batch_size 2 (conf/sample/default.yaml:1-2), graph replay
        blk["b2_latency_T1000_B2_N30_L6"] = dict(full(2, timeit(lambda: host(m6.sample_from_masks(nm2, None, None))), 6),
                                                 launch="hipGraph replay (cached)")
        # ... and the whole shipped job, `sample_batches(batch_size=2, num_batches=16)` with sizes drawn from the GEOM histogram
        # (diffusion_qm9.py:397-436): the 32 independent molecules run as one device batch (DiffusionQM9.merge_batches),
        # bit-identical to the reference's loop order of 16 device batches, which is timed beside it in fp32
        def job(merge):
            m6.merge_batches = merge
            torch.manual_seed(2022)
            return m6.sample_batches(2, 16, dev)
        dt_job = timeit(lambda: job(4096))
        blk["shipped_job_16_batches_of_2_L6"] = {"s_per_job": round(dt_job, 4), "molecules_per_s": round(32 / dt_job, 2), "timesteps": 1000,
                                                 "how": "one device batch of 32 (merge_batches)"}
        if prec == "fp32":
            dt_loop = timeit(lambda: job(0))
            blk["shipped_job_16_batches_of_2_L6"].update(s_per_job_as_16_device_batches=round(dt_loop, 4),
                                                         speedup_from_merging=round(dt_loop / dt_job, 2))
        m6.merge_batches = 4096
        # a GEOM-sized job of several batches (config 3's sizes: 4 batches of 256): merged into one device batch (up to
        # DiffusionQM9.merge_edges = 900,000 edges, four headline batches) the edge kernel runs at its full tile rate
        def geom_job(merge):
            m6.merge_batches = merge
            torch.manual_seed(2022)
            return m6.sample_batches(256, 4, dev)
        dt_g = timeit(lambda: geom_job(4096))
        blk["geom_job_4_batches_of_256_L6"] = {"s_per_job": round(dt_g, 4), "molecules_per_s": round(1024 / dt_g, 2), "timesteps": 1000,
                                               "how": "merged device batches (merge_batches / merge_edges)"}
        if prec == "fp32":
            dt_gl = timeit(lambda: geom_job(0))
            blk["geom_job_4_batches_of_256_L6"].update(s_per_job_as_4_device_batches=round(dt_gl, 4), speedup_from_merging=round(dt_gl / dt_g, 2))
        m6.merge_batches = 4096
        blk["headline_L9_B256_N30"] = short(256, timeit(lambda: m9s.sample_from_masks(nm256, None, None), reps=2), Ts)
        # graph size of a pocket-conditioned job (30 fragments + 170 pocket residues in one graph, diffusion_qm9.py:362-371)
        blk["pocket_sized_B32_N200_L6"] = short(32, timeit(lambda: m6s.sample_from_masks(nmp, None, None), reps=2), Ts)
        out[DTYPE[prec]] = blk
    return out


def next_rows(dev) -> dict:
    """Timings of the rows behind the hot path: one training step of the same EGNN (row 2: loss
    forward + backward through the HIP edge-layer kernels + AdamW, exact fp32, synthetic batch) and one stage-2 E_GCL
    layer forward on a beam-sized dense batch (row 4).  Parity of both is the GPU test tier's job (tests/test_gpu_training.py,
    tests/test_stage2.py); these are single-GPU figures, reported for completeness, not part of `value`."""
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.stage2 import E_GCL, synthetic_egcl_state_dict
    from hierdiff_amd.weights import synthetic_state_dict
    out = {}
    H, L, N = 256, 6, 30
    m = DiffusionQM9(default_config(hidden_nf=H, n_layers=L))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_state_dict(9, 0, H, L, 2, True, 0, 0.5).items()})
    m = m.to(dev).train()
    from hierdiff_amd.trainer import configure_optimizers
    opt, _ = configure_optimizers(m, lr=1e-4)          # the package's own factory: AdamW with the reference's values, fused on the GPU
    for B in (256, 64):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, N, 3, generator=g)
        x = x - x.mean(1, keepdim=True)
        h = torch.cat([torch.randint(0, 5, (B, N, 5), generator=g).float(), torch.randn(B, N, 3, generator=g)], 2)
        batch = {"positions": x.to(dev), "atom_mask": torch.ones(B, N, 1, dtype=torch.bool, device=dev),
                 "edge_mask": (~torch.eye(N, dtype=torch.bool))[None].expand(B, N, N).contiguous().to(dev),
                 "node_feature": h.to(dev)}

        def step():
            opt.zero_grad(set_to_none=True)
            loss = m.training_step(batch, 0)
            loss.backward()
            opt.step()

        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 5
        with torch.no_grad():
            m.forward(batch)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                m.forward(batch)
            torch.cuda.synchronize(dev)
            dv = (time.perf_counter() - t0) / 5
        out[f"training_step_B{B}_N30_L6_f32"] = {"ms_per_step": round(dt * 1e3, 2), "molecules_per_s": round(B / dt, 1),
                                                 "loss_value_no_grad_ms": round(dv * 1e3, 2),
                                                 "what": "DiffusionQM9.training_step + backward + AdamW.step, the same batch every step "
                                                         "(its topology is cached); every kernel exact fp32"}
        # opt-in mixed mode (dynamics.training_precision = "fp16x3"): the edge layer's contraction sites in the two-way FP16 split, on top
        # of the kept second-layer pre-activations
        m.dynamics.training_precision = "fp16x3"
        for _ in range(2):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize(dev)
        d3 = (time.perf_counter() - t0) / 5
        m.dynamics.training_precision = "fp32"
        out[f"training_step_B{B}_N30_L6_fp16x3_contractions"] = {
            "ms_per_step": round(d3 * 1e3, 2), "molecules_per_s": round(B / d3, 1),
            "what": "training_precision = 'fp16x3' (precision 3 + hd_dw2_f16; layers of a batch too small to keep pre2 run in exact fp32): "
                    "gradients within 2e-6 of the exact-fp32 step's (tests/test_gpu_training.py)"}
        # what a real training loop sees: NEW masks every step, i.e. one topology build per step inside the timed region.  Same
        # WORK in the rows below: one multiset of ragged sizes (12 .. 30 nodes, mean 21), either the same batch every step
        # (topology cached) or the sizes permuted over the batch positions every step (never-seen masks, identical edge / node
        # counts) - the latter twice: HOST batches staged one step ahead the way trainer.fit_epoch does (stage_batch: layout from
        # the host masks, async upload into a pooled arena, no host wait), and batches whose masks exist on the DEVICE only (the
        # fallback: one device-to-host mask copy per step, which waits for everything queued).
        rng = np.random.Generator(np.random.PCG64(B))
        sizes0 = rng.integers(12, N + 1, B)

        def ragged(perm):
            sizes = torch.from_numpy(sizes0[perm])
            nmk = (torch.arange(N)[None, :] < sizes[:, None])
            emk = nmk[:, :, None] & nmk[:, None, :] & ~torch.eye(N, dtype=torch.bool)[None]
            xk = torch.randn(B, N, 3, generator=g) * nmk[..., None]
            xk = xk - (xk.sum(1, keepdim=True) / sizes.view(-1, 1, 1)) * nmk[..., None]
            return {"positions": xk, "atom_mask": nmk[..., None], "edge_mask": emk, "node_feature": h * nmk[..., None]}

        # 22 never-seen batches per variant: the first 14 steps are untimed (they fill the two 8-entry topology caches and the
        # arena pool - a loop's steady state recycles arenas, its first steps allocate them), the last 8 are timed
        on_dev = lambda bt: {k: v.to(dev) for k, v in bt.items()}
        NB, NW_ = 22, 14
        fresh_host = [ragged(rng.permutation(B)) for _ in range(NB)]
        fresh = [on_dev(ragged(rng.permutation(B))) for _ in range(NB)]
        same = on_dev(ragged(np.arange(B)))
        torch.cuda.synchronize(dev)

        def step_on(bt):
            opt.zero_grad(set_to_none=True)
            loss = m.training_step(bt, 0)
            loss.backward()
            opt.step()

        def timed(batches, warm=2):
            for bt in batches[:warm]:
                step_on(bt)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for bt in batches[warm:]:
                step_on(bt)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / (len(batches) - warm)

        def timed_staged(batches):                      # the loop of trainer.fit_epoch: stage k+1 behind the launch of step k
            cur = m.stage_batch(batches[0], dev)
            t0 = None
            for k in range(len(batches)):
                if k == NW_:
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                step_on(cur)
                if k + 1 < len(batches):
                    cur = m.stage_batch(batches[k + 1], dev)
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / (len(batches) - NW_)

        dc = timed([same] * 10)
        df = timed(fresh, NW_)
        ds = timed_staged(fresh_host)
        out[f"training_step_B{B}_N30_L6_f32_ragged_cached_masks"] = {
            "ms_per_step": round(dc * 1e3, 2), "molecules_per_s": round(B / dc, 1), "mean_nodes": round(float(sizes0.mean()), 1),
            "what": "ragged sizes 12..30, the same batch every step (topology cached)"}
        out[f"training_step_B{B}_N30_L6_f32_fresh_masks"] = {
            "ms_per_step": round(ds * 1e3, 2), "molecules_per_s": round(B / ds, 1), "mean_nodes": round(float(sizes0.mean()), 1),
            "topology_build_ms_per_step": round((ds - dc) * 1e3, 2),
            "device_only_masks_ms_per_step": round(df * 1e3, 2),
            "what": "the SAME sizes permuted over the batch positions every step: masks never seen before (one topology build per "
                    "step inside the timed region), identical node / edge counts as the cached row above; host batches staged one "
                    "step ahead (DiffusionQM9.stage_batch, the loop of trainer.fit_epoch; host-to-device copies of the batch "
                    "included); device_only_masks = the fallback for masks that exist on the device only"}
    del m, opt
    # stage 2: gcl_full layer of edge_denoise.py:35-43 (H-wide edge features, attention, edge update), bs graphs of n nodes
    bs, n = 24, 12
    ar = torch.arange(n)
    row = (ar.repeat_interleave(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n)).to(dev)
    col = (ar.repeat(n).repeat(bs) + (torch.arange(bs) * n).repeat_interleave(n * n)).to(dev)
    g = torch.Generator().manual_seed(1)
    hh = torch.randn(bs * n, H, generator=g).to(dev)
    xx = torch.randn(bs * n, 3, generator=g).to(dev)
    ea = torch.randn(row.numel(), H, generator=g).to(dev)
    nmask = torch.ones(bs * n, 1, device=dev)
    emask = (row != col).float().unsqueeze(1)
    lay = E_GCL(H, H, H, edges_in_d=H, attention=True, tanh=True, coords_range=30, edge_update=True)
    lay.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_egcl_state_dict(H, H, 0, True, True, 40, coord_gain=0.3).items()})
    lay = lay.to(dev)
    for _ in range(3):
        lay(hh, [row, col], xx, edge_attr=ea, node_mask=nmask, edge_mask=emask)
    lay._frozen = True          # as inside Edge_denoise: parameters verified once per model call (key + content digest), not per layer
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(20):
        lay(hh, [row, col], xx, edge_attr=ea, node_mask=nmask, edge_mask=emask)
    torch.cuda.synchronize(dev)
    t_layer = (time.perf_counter() - t0) / 20
    del lay
    # the stage-2 model around the layer: one autoregressive growth step (Edge_denoise.sample_AR) for a beam of 24 half-grown
    # 12-node trees at the production width (conf/model/edge_denoise.yaml: H = 256, vocabulary 781, 3 + 3 layers)
    from hierdiff_amd.edge_denoise import Edge_denoise, synthetic_edge_denoise_state_dict
    kw = dict(vocab_size=781, in_node_nf=8, hidden_nf=H, out_node_nf=780, context_nf=0)
    ed = Edge_denoise(array_dict=None, full_softmax=True, **kw)
    ed.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synthetic_edge_denoise_state_dict(3, **kw).items()})
    ed = ed.to(dev)
    rng = np.random.Generator(np.random.PCG64(5))
    adj = torch.zeros(bs, n, n)
    for b in range(bs):
        for v in range(1, 6):                                   # six placed nodes: a random tree over nodes 0..5
            p = int(rng.integers(0, v))
            adj[b, v, p] = adj[b, p, v] = 1
    feat = torch.from_numpy(rng.standard_normal((bs, n, 10)).astype(np.float32))
    feat[:, :, 9] = torch.from_numpy(rng.integers(0, 780, (bs, n)).astype(np.float32))
    beam = {'node_feat': [feat.to(dev), torch.ones(bs, n, 10, device=dev)], 'node_pos': (torch.randn(bs, n, 3, generator=g) * 1.5).to(dev),
            'search_adj_matrix': adj.to(dev), 'edge_mask': (1 - torch.eye(n))[None].expand(bs, n, n).contiguous().to(dev)}
    for _ in range(2):
        ed.sample_AR(beam)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(5):
        ed.sample_AR(beam)
    torch.cuda.synchronize(dev)
    out["stage2_Edge_denoise_sample_AR_bs24_n12_H256"] = {
        "ms_per_step": round((time.perf_counter() - t1) / 5 * 1e3, 2), "layers": "3 gcl_full + 3 gcl_focal + gcl_edge / gcl_denoise walks",
        "what": "one growth step of the beam incl. the host-side tree bookkeeping (breadth-first layers, argmax, adjacency)"}
    out["stage2_E_GCL_layer_bs24_n12_H256"] = {"ms_per_layer_forward": round(t_layer * 1e3, 3),
                                               "nodes": bs * n, "edges": int(row.numel()),
                                               "what": "gcl_full layer (edge features + attention + edge update), exact fp32"}
    return out


def launcher_command(n_gpus: int, argv, port: int) -> list:
    """The command line `python bench.py --gpus N ...` re-executes itself as when no launcher environment is present:
    torch.distributed.run, one node, one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    rest = [a for a in argv if a != "--force-launcher"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + rest


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_gpus: int, argv) -> int:
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = launcher_command(n_gpus, argv, free_port())
    print("bench.py: no launcher environment, starting " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def driver_visible_order(out: dict) -> dict:
    """Key order of the JSON line.  The driver's record keeps the last ~8 KB of the line, so the
    blocks this tier credits go LAST: the rows outside the hot path (training, stage 2) first,
    then `configs` with the exact-fp32 block as its last entry, the fp16x3 headline block, and finally the contract's own
    keys with `roofline` and `cpu_baseline`.  A JSON object is unordered for every parser; this is about the truncated copy."""
    first = ["next_rows", "configs", "fp16x3"]
    ordered = {k: out[k] for k in first if k in out}
    if isinstance(ordered.get("configs"), dict):
        c = ordered["configs"]
        inner = [k for k in c if k not in ("fp16x3", "f32")] + [k for k in ("fp16x3", "f32") if k in c]
        ordered["configs"] = {k: c[k] for k in inner}
    last = ["config", "roofline", "cpu_baseline"]
    for k, v in out.items():
        if k not in ordered and k not in last:
            ordered[k] = v
    for k in last:
        if k in out:
            ordered[k] = out[k]
    return ordered


NOTES = {"fp32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's arithmetic",
         "fp16x3": "fp32-accurate: per-edge H x H contraction on a two-way fp16 split (22 significant bits, operands ranged by exact "
                   "powers of two), 3 fp16 MFMAs per product, fp32 accumulate; node update in the same two-piece fp16 arithmetic "
                   "from width 128 up (k_node<..., F16>), exact fp32 node kernels below"}


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
    ap.add_argument("--precision", choices=["all", "fp32", "fp16x3"], default="all",
                    help="'all' (default): headline = exact fp32, plus the fp16x3 sibling block; a single mode "
                         "times only that mode (profiling runs)")
    ap.add_argument("--graph", action="store_true", help="replay each diffusion step from a captured hipGraph")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket the edge-kernel launches of every k-th forward with HIP events")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs block (BASELINE configs 2, 3, 5, B=2)")
    ap.add_argument("--config-timesteps", type=int, default=50)
    ap.add_argument("--no-dist", action="store_true",
                    help="under the launcher at world size 1: do not initialise torch.distributed")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket edge-kernel launches with HIP events in the timed region")
    ap.add_argument("--force-launcher", action="store_true",
                    help="start the ranks through torch.distributed.run even for --gpus 1 (the N > 1 code path on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if (args.gpus > 1 or args.force_launcher) and "RANK" not in os.environ:
        # called plainly (`python bench.py --gpus N`): start the ranks ourselves, one process per GPU, exactly as the
        # driver's launcher form does; rank 0's JSON line is this process's output and its exit code ours
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # Under the launcher (torch.distributed.run exports RANK / WORLD_SIZE / MASTER_*) the process group is ALWAYS
    # initialised, also at world size 1: RCCL is loaded, ncclBroadcast / barrier / all-gather are issued - the same code
    # path as N = 8, so the single-GPU test tier covers it (tests/test_gpu_configs.py).  A plain `python bench.py` has no
    # rendezvous environment and runs without torch.distributed.
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ and not args.no_dist):
        import torch.distributed as dist  # type: ignore
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    H, L, B, N, T = args.hidden, args.layers, args.batch, args.nodes, args.timesteps
    model = build_model(H, L, T, dev, rank, world, dist_on=dist is not None)
    modes = ["fp32", "fp16x3"] if args.precision == "all" else [args.precision]
    blocks = {p: timed_headline(model, p, args, dev, rank, world, dist) for p in modes}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    head = modes[0]
    hb = blocks[head]
    n_fwd = T + 1
    out = {
        "metric": "sampled molecules/sec (1000 diffusion steps, B=256, N=30)",
        "value": hb["value"], "unit": "molecules/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": hb["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[head], "data": "synthetic",
        "config": {"workload": f"DiffusionQM9.sample: {T}-step reverse diffusion + decode ({n_fwd} EGNN forwards), "
                               f"B={B} per GPU, N={N} all valid, H={H}, L={L}, S=2",
                   "batch_per_gpu": B, "n_nodes": N, "hidden_nf": H, "n_layers": L, "timesteps": T,
                   "precision": head,
                   "precision_note": NOTES[head],
                   "launch": hb["launch"],
                   "parallelism": f"{world} independent shards, RCCL weight broadcast only"},
        "ms_per_forward": hb["ms_per_forward"], "model_tflops": hb["model_tflops"],
        "frac_end_to_end": hb["frac_end_to_end"],
        "elapsed_max_s": hb["elapsed_max_s"], "rank_elapsed_s": hb["rank_elapsed_s"],
    }
    if dist is not None:
        out["rccl_ranks"] = world
        out["rccl"] = {"backend": dist.get_backend(), "broadcast_elements": getattr(model, "rccl_broadcast_elements", 0),
                       "collectives": "1 weight broadcast at start-up; barrier + all-gather of the per-rank times around "
                                      "the timed region; none on the data path"}
    if "roofline" in hb:
        out["roofline"] = hb["roofline"]
    for mode in ("fp16x3",):
        if mode in blocks and head != mode:
            sib = dict(blocks[mode])
            sib["precision_note"] = "opt-in mode: " + NOTES[mode] + "; same K steps, same workload, same process as the headline"
            if world == 1 and "fp32" in blocks:
                sib.update(precision_gap(model, args, dev, mode))
            out[mode] = sib
    if world == 1 and not args.no_configs:
        out["configs"] = other_configs(args, dev)
        out["next_rows"] = next_rows(dev)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(H, L, B, N, T)
    print(json.dumps(driver_visible_order(out)), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(H: int, L: int, B: int, N: int, T: int) -> dict:
    """The oracle (a structural restatement of the reference's PyTorch-CPU op sequence, pinned to the reference by
    the golden vectors) timed on this host at the headline batch: 1 warm-up + K=5 EGNN dynamics forwards at B=256
    (BASELINE.md section 3), extrapolated to a full sample (T+1 forwards)."""
    from hierdiff_amd.weights import synthetic_state_dict
    from oracle import egnn_oracle as orc
    logical = os.cpu_count() or 1
    try:
        logical = len(os.sched_getaffinity(0))
    except Exception:
        pass
    phys = logical
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or logical
    except Exception:
        pass
    sd = orc.as_torch_sd(synthetic_state_dict(9, 0, H, L, 2, True, seed=0, coord_gain=1.0))
    cfg = orc.DynCfg(hidden_nf=H, n_layers=L)
    # torch's intra-op pool scales poorly on big dual-socket hosts (128 threads measured 3x SLOWER than 16 on this op
    # mix): probe a few pool sizes with one forward each at a quarter batch and time the fastest - the conservative
    # (strongest) CPU baseline.
    allc = min(phys, logical)
    cands = sorted({c for c in (8, 16, 32, 64, allc) if c <= logical})
    Bq = max(1, B // 4)
    probe = {c: orc.time_cpu_forward(sd, cfg, Bq, N, 8, 1, c, time.perf_counter) for c in cands}
    threads = min(probe, key=probe.get)
    K = 3
    sec = orc.time_cpu_forward(sd, cfg, B, N, 8, K, threads, time.perf_counter)
    # BASELINE.md section 3 asks for "all physical cores": the same oracle with one torch thread per physical core, timed on
    # the quarter batch of the probe (K = 2 after its warm-up) - on big hosts this is the SLOWER configuration
    sec_all = sec if threads == allc else orc.time_cpu_forward(sd, cfg, Bq, N, 8, 2, allc, time.perf_counter)
    b_all = B if threads == allc else Bq
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except Exception:
        pass
    return {"value": round(B / ((T + 1) * sec), 5), "unit": "molecules/s", "cores": threads, "host_cores": phys,
            "host_logical_cpus": logical, "host_cpu": model, "kind": "port",
            "sample": f"K={K} EGNN dynamics forwards after 1 warm-up at B={B}, N={N}, H={H}, L={L}, fp32, {threads} torch "
                      f"threads (fastest of {cands} in a 1-forward probe at B={Bq}: "
                      f"{ {c: round(v, 2) for c, v in probe.items()} } s); {sec:.3f} s/forward, extrapolated x{T + 1} "
                      "forwards per batch",
            "s_per_forward": round(sec, 4),
            "all_physical_cores": {"value": round(b_all / ((T + 1) * sec_all), 5), "unit": "molecules/s", "cores": allc,
                                   "sample": f"K=2 forwards after 1 warm-up at B={b_all}, {allc} torch threads: "
                                             f"{sec_all:.3f} s/forward, extrapolated x{T + 1}",
                                   "s_per_forward": round(sec_all, 4)}}


if __name__ == "__main__":
    main()
