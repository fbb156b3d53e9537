"""CPU tier: the N>1 path (world_size 2, gloo): rank 0's parameters reach every rank with one broadcast,
global sample ids are split without gaps or overlap, and the counter RNG depends only on the global id."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hierdiff_amd.sharding import (broadcast_model_weights, pack_parameters, shard_sample_ids, shard_sizes,
                                   unpack_parameters)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hierdiff_amd import DiffusionQM9, _lib, default_config
    from hierdiff_amd.weights import synthetic_state_dict
    torch.manual_seed(100 + rank)                       # ranks start from different random inits
    model = DiffusionQM9(default_config(hidden_nf=32, n_layers=2))
    if rank == 0:
        sd = synthetic_state_dict(9, 0, 32, 2, 2, True, 3)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    n = broadcast_model_weights(model, src=0)
    flat = pack_parameters(model)
    # every rank now holds rank 0's synthetic weights
    ref = synthetic_state_dict(9, 0, 32, 2, 2, True, 3)
    for k, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), ref[k]), k
    # this rank's shard of 11 global samples starting at id 1000, and its share of one noise draw
    start, count = shard_sample_ids(1000, 11, rank, world)
    lib = _lib.load()
    draws = np.array([[lib.hd_philox_normal_host(2022, start + b, 5, i) for i in range(8)] for b in range(count)],
                     dtype=np.float32)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), start=start, count=count, draws=draws, n=n,
             checksum=float(flat.double().sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_broadcast_and_sharding(tmp_path):
    from hierdiff_amd import build
    build.build(verbose=False)
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f"rank{k}.npz")) for k in range(world)]
    assert r[0]["checksum"] == r[1]["checksum"] and r[0]["n"] == r[1]["n"]
    assert int(r[0]["start"]) == 1000 and int(r[0]["count"]) == 6
    assert int(r[1]["start"]) == 1006 and int(r[1]["count"]) == 5
    # single-process reference for the same global ids
    from hierdiff_amd import _lib
    lib = _lib.load()
    full = np.array([[lib.hd_philox_normal_host(2022, 1000 + b, 5, i) for i in range(8)] for b in range(11)],
                    dtype=np.float32)
    assert np.array_equal(np.concatenate([r[0]["draws"], r[1]["draws"]]), full)


@pytest.mark.parametrize("total,world", [(2048, 8), (10, 4), (3, 8), (0, 2), (257, 2)])
def test_shard_partition_is_exact(total, world):
    spans = [shard_sample_ids(7, total, r, world) for r in range(world)]
    assert sum(c for _, c in spans) == total and shard_sizes(total, world) == [c for _, c in spans]
    nxt = 7
    for s, c in spans:
        assert s == nxt and c >= 0
        nxt += c
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_pack_unpack_roundtrip():
    m = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.BatchNorm1d(5))
    flat = pack_parameters(m)
    m2 = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.BatchNorm1d(5))
    unpack_parameters(m2, flat)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.float(), b.float()), k
    with pytest.raises(ValueError):
        unpack_parameters(m2, flat[:-1])


def _grad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hierdiff_amd import DiffusionQM9, default_config
    from hierdiff_amd.sharding import allreduce_gradients, broadcast_model_weights
    torch.manual_seed(7 + rank)
    model = DiffusionQM9(default_config(hidden_nf=32, n_layers=1))
    broadcast_model_weights(model, src=0)
    # rank-dependent synthetic gradients (the HIP backward needs a GPU; the collective does not care where they came
    # from); one parameter is left without a gradient on rank 1 only
    g = torch.Generator().manual_seed(1000 + rank)
    params = [p for p in model.parameters() if p.requires_grad]
    for k, p in enumerate(params):
        if rank == 1 and k == 3:
            continue
        p.grad = torch.randn(p.shape, generator=g)
    n = allreduce_gradients(model)
    torch.save({"n": n, "grads": [p.grad.clone() for p in params]}, os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gradient_allreduce(tmp_path):
    """The training path's only collective: one flat all-reduce averaging every gradient (DDP, conf/trainer/default.yaml:2-3)."""
    world, port = 2, _free_port()
    mp.spawn(_grad_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"g{k}.pt") for k in range(world)]
    assert r[0]["n"] == r[1]["n"] > 0
    from hierdiff_amd import DiffusionQM9, default_config
    shapes = [p.shape for p in DiffusionQM9(default_config(hidden_nf=32, n_layers=1)).parameters() if p.requires_grad]
    gens = [torch.Generator().manual_seed(1000 + k) for k in range(world)]
    for k, shp in enumerate(shapes):
        a = torch.randn(shp, generator=gens[0])
        b = torch.zeros(shp) if k == 3 else torch.randn(shp, generator=gens[1])
        want = (a + b) / 2
        assert torch.allclose(r[0]["grads"][k], want, atol=1e-7) and torch.equal(r[0]["grads"][k], r[1]["grads"][k]), k


class _ToyModel(torch.nn.Module):
    """Stands in for DiffusionQM9 in the CPU tier (the real training_step needs a GPU): same `training_step(batch, idx) -> loss`."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(6, 3)

    def training_step(self, batch, batch_idx=0):
        return ((self.lin(batch["x"]) - batch["y"]) ** 2).mean()


def _step_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hierdiff_amd.trainer import configure_optimizers, fit_epoch
    torch.manual_seed(5)                                   # identical initial weights on every rank (what the broadcast gives)
    model = _ToyModel()
    opt, sched = configure_optimizers(model)
    g = torch.Generator().manual_seed(40 + rank)          # every rank its own shard of the global batch
    batches = [{"x": torch.randn(8, 6, generator=g) * 30, "y": torch.randn(8, 3, generator=g)} for _ in range(3)]
    log = fit_epoch(model, batches, opt, sched, clip_val=2.0)
    torch.save({"w": model.lin.weight.detach().clone(), "b": model.lin.bias.detach().clone(), "log": log,
                "lr": opt.param_groups[0]["lr"]}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.autograd          # (tests run under torch.no_grad() unless marked: tests/conftest.py)
def test_world2_ddp_step_matches_single_process_on_the_global_batch(tmp_path):
    """hierdiff_amd.trainer: loss.backward -> one flat gradient all-reduce (mean) -> clip_grad_norm_(2) -> AdamW.step, the
    reference trainer's per-batch sequence (conf/trainer/default.yaml, conf/optim/adamw.yaml).  Two ranks on two shards end with
    the same weights as one process that averages the two shards' gradients itself."""
    world, port = 2, _free_port()
    mp.spawn(_step_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"s{k}.pt") for k in range(world)]
    assert torch.equal(r[0]["w"], r[1]["w"]) and torch.equal(r[0]["b"], r[1]["b"])
    assert all(a["grad_norm"] == b["grad_norm"] for a, b in zip(r[0]["log"], r[1]["log"]))     # norm of the AVERAGED gradient
    assert max(a["grad_norm"] for a in r[0]["log"]) > 2.0, "the test must exercise the clipping"
    # single-process restatement
    from hierdiff_amd.trainer import configure_optimizers
    torch.manual_seed(5)
    model = _ToyModel()
    opt, sched = configure_optimizers(model)
    assert opt.defaults["lr"] == 4.0e-4 and opt.defaults["weight_decay"] == 4.0e-8 and sched.step_size == 15 and sched.gamma == 0.1
    gens = [torch.Generator().manual_seed(40 + k) for k in range(world)]
    for _ in range(3):
        shards = [{"x": torch.randn(8, 6, generator=g) * 30, "y": torch.randn(8, 3, generator=g)} for g in gens]
        opt.zero_grad(set_to_none=True)
        grads = []
        for sh in shards:
            model.zero_grad(set_to_none=True)
            model.training_step(sh).backward()
            grads.append([p.grad.clone() for p in model.parameters()])
        for p, ga, gb in zip(model.parameters(), *grads):
            p.grad = (ga + gb) / 2
        torch.nn.utils.clip_grad_norm_(list(model.parameters()), 2.0)
        opt.step()
    assert torch.allclose(model.lin.weight, r[0]["w"], atol=1e-7) and torch.allclose(model.lin.bias, r[0]["b"], atol=1e-7)
    assert r[0]["lr"] == 4.0e-4                       # StepLR: unchanged after one epoch of fifteen


def test_bench_self_launch_command_line(monkeypatch):
    """`python bench.py --gpus 2` with no RANK in the environment must not exit on argument handling: it re-executes itself
    under torch.distributed.run with one rank per GPU on 127.0.0.1 (bench.launcher_command / self_launch); under the
    launcher (RANK set) it does not launch again."""
    import importlib
    import os
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    bench = importlib.import_module("bench")
    cmd = bench.launcher_command(2, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--force-launcher"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(repo, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]          # the flag that forced the launch is dropped
    # main(): --gpus 2 without a launcher environment goes through self_launch with the untouched argument list
    seen = {}
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: seen.update(n=n, argv=list(argv)) or 0)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"n": 2, "argv": ["--gpus", "2", "--steps", "3"]}
    assert 1024 < bench.free_port() < 65536


def test_bench_line_keeps_the_creditable_blocks_in_the_drivers_tail():
    """The driver's record keeps the last ~8 KB of bench.py's JSON line: `driver_visible_order` must put the
    exact-fp32 configs block, the fp16x3 blocks, the contract's keys, `roofline` and `cpu_baseline` there, lose nothing and stay
    valid JSON with the same content."""
    import json
    import bench
    blk = lambda n: {f"case{i}": {"molecules_per_s": 1.0 * i, "ms_per_forward": 2.0, "what": "x" * 40} for i in range(n)}
    line = {"metric": "m", "value": 1.0, "unit": "molecules/s", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 5.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w"}, "roofline": {"frac": 0.7, "pad": "r" * 600},
            "fp16x3": {"value": 2.0, "pad": "p" * 1500},
            "configs": {"note": "n", "f32": blk(10), "fp16x3": blk(10)},
            "next_rows": {f"row{i}": {"ms_per_step": 1.0, "what": "y" * 300} for i in range(10)},
            "cpu_baseline": {"value": 0.03, "sample": "s" * 400}}
    out = bench.driver_visible_order(line)
    assert out == line and list(out)[-3:] == ["config", "roofline", "cpu_baseline"]
    assert list(out["configs"])[-2:] == ["fp16x3", "f32"]
    text = json.dumps(out)
    assert json.loads(text) == line
    tail = text[-8000:]
    i_f32 = tail.find('"f32": {"case0"')
    assert i_f32 >= 0, "the fp32 configs block fell out of the tail"
    assert '"fp16x3": {"value"' in tail and '"metric"' in tail and '"cpu_baseline"' in tail and '"roofline"' in tail
    assert '"next_rows"' not in tail                      # the rows outside the hot path go first


# ----------------------------------------------------------------------------- round 6: environment, overlapped buckets, epoch-end hooks

@pytest.mark.parametrize("entry", ["import hierdiff_amd", "import bench", "import hierdiff_amd.sharding"])
def test_every_entry_point_sets_dmabuf_ipc_before_hip_loads(entry):
    """RCCL between the per-GPU ranks needs HSA_ENABLE_IPC_MODE_LEGACY=0 on the MI355X boxes' driver.  A rank started by ANY launcher
    (not only bench.py's own) must have it: importing the package or bench.py sets it in-process; a caller's own value is kept."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os; assert 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ; " + entry +
            "; print('IPC=' + os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])")
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    out = subprocess.run([sys.executable, "-c", code], cwd=repo, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "IPC=0" in out.stdout
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"
    code = entry + "; import os; print('IPC=' + os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])"
    out = subprocess.run([sys.executable, "-c", code], cwd=repo, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "IPC=1" in out.stdout


class _Block(torch.nn.Module):
    def __init__(self, w):
        super().__init__()
        self.a = torch.nn.Linear(w, w)
        self.b = torch.nn.Linear(w, w)

    def forward(self, h):
        return h + self.b(torch.tanh(self.a(h)))


class _BlockedToy(torch.nn.Module):
    """Parameter names shaped like the product's (`egnn.e_block_<i>.*` + embedding / output layer / an unused parameter), differentiated
    on the CPU: blocks finish back to front in backward, as the EGNN's do."""

    def __init__(self, w=5, blocks=3):
        super().__init__()
        self.egnn = torch.nn.Module()
        self.egnn.embedding = torch.nn.Linear(4, w)
        for i in range(blocks):
            self.egnn.add_module(f"e_block_{i}", _Block(w))
        self.egnn.embedding_out = torch.nn.Linear(w, 2)
        self.unused = torch.nn.Parameter(torch.ones(3))
        self.blocks = blocks

    def training_step(self, batch, batch_idx=0):
        h = self.egnn.embedding(batch["x"])
        for i in range(self.blocks):
            h = getattr(self.egnn, f"e_block_{i}")(h)
        return ((self.egnn.embedding_out(h) - batch["y"]) ** 2).mean()


def _bucket_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"        # set by importing the package inside this rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hierdiff_amd.sharding import GradientBuckets, allreduce_gradients
    res = {}
    for mode in ("flat", "buckets"):
        torch.manual_seed(5)
        model = _BlockedToy()
        g = torch.Generator().manual_seed(70 + rank)
        batch = {"x": torch.randn(16, 4, generator=g), "y": torch.randn(16, 2, generator=g)}
        if mode == "buckets":
            bk = GradientBuckets(model)
            model.training_step(batch).backward()
            early = list(bk.launch_order)             # what was already on the wire when backward returned
            n = bk.finish()
            res["early"], res["order"], res["n_b"] = early, list(bk.launch_order), n
            # a second step through the same object: state is reset, same result for the same batch
            model.zero_grad(set_to_none=True)
            model.training_step(batch).backward()
            bk.finish()
            bk.remove()
        else:
            model.training_step(batch).backward()
            res["n_f"] = allreduce_gradients(model)
        res[mode] = [p.grad.clone() for p in model.parameters()]
    # gradient accumulation: two backward passes, ONE finish() - the buckets sent during the first pass are stale and must be re-sent
    acc = {}
    for mode in ("flat", "buckets"):
        torch.manual_seed(5)
        model = _BlockedToy()
        bk = GradientBuckets(model) if mode == "buckets" else None
        for k in range(2):
            g = torch.Generator().manual_seed(90 + 10 * k + rank)
            model.training_step({"x": torch.randn(16, 4, generator=g), "y": torch.randn(16, 2, generator=g)}).backward()
        if bk is not None:
            bk.finish(); bk.remove()
        else:
            allreduce_gradients(model)
        acc[mode] = [p.grad.clone() for p in model.parameters()]
    res["acc_equal"] = all(torch.equal(a, b) for a, b in zip(acc["flat"], acc["buckets"]))
    torch.save(res, os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.autograd
def test_world2_overlapped_buckets_equal_the_flat_allreduce(tmp_path):
    """GradientBuckets (one all-reduce per EGNN block, issued from backward hooks back to front) gives bit for bit the gradients of the
    single flat all-reduce, on both ranks, with a parameter that never receives a gradient riding along as zeros."""
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"b{k}.pt") for k in range(world)]
    for k in range(world):
        assert r[k]["n_b"] == r[k]["n_f"] > 0
        assert r[k]["early"] == [0, 1, 2], "the three block buckets (highest block first) must be launched DURING backward"
        assert r[k]["order"] == [0, 1, 2, 3]           # the remainder (embedding, output layer, unused) at finish()
        for a, b in zip(r[k]["flat"], r[k]["buckets"]):
            assert torch.equal(a, b)
    for a, b in zip(r[0]["buckets"], r[1]["buckets"]):
        assert torch.equal(a, b)
    assert r[0]["acc_equal"] and r[1]["acc_equal"], "two backward passes before one finish() must equal the flat all-reduce"
    assert torch.equal(r[0]["buckets"][-1], torch.zeros(3)) or torch.equal(r[0]["buckets"][0], torch.zeros(3))


def _epoch_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hierdiff_amd import DiffusionQM9, default_config
    model = DiffusionQM9(default_config(hidden_nf=32, n_layers=1))
    steps = [{"loss": torch.tensor(1.0 + rank + 10 * k)} for k in range(3)]
    model.validation_epoch_end(steps)
    model.test_epoch_end(steps)
    gathered = model._gather_result([{"loss": torch.tensor(float(rank))}, {"loss": torch.tensor(5.0)}])
    torch.save({"logged": {k: float(v) for k, v in model.logged.items()}, "gathered": gathered["loss"]}, os.path.join(out_dir, f"e{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_epoch_end_hooks_and_configure_optimizers(tmp_path):
    """The Lightning-side hooks of the reference module (diffusion_qm9.py:753-801, 871-879) exist on DiffusionQM9 and work without
    Lightning: steps are joined, ranks gathered, the mean logged (test metric on rank 0 only); configure_optimizers returns
    ([optimizer], [scheduler]) from cfg.optim / cfg.scheduler or the reference's shipped values."""
    from hierdiff_amd import DiffusionQM9, default_config
    model = DiffusionQM9(default_config(hidden_nf=32, n_layers=1))
    # one process: vector-valued and scalar step outputs
    out = model._gather_result([{"loss": torch.tensor([1.0, 2.0])}, {"loss": torch.tensor([6.0])}])
    assert torch.equal(out["loss"], torch.tensor([1.0, 2.0, 6.0]))
    model.validation_epoch_end([{"loss": torch.tensor(2.0)}, {"loss": torch.tensor(4.0)}])
    assert float(model.logged["val_loss"]) == 3.0
    model.test_epoch_end([{"loss": torch.tensor(2.0)}, {"loss": torch.tensor(8.0)}])
    assert float(model.logged["test/ppl"]) == 5.0
    opts, scheds = model.configure_optimizers()
    assert isinstance(opts, list) and isinstance(scheds, list) and len(opts) == len(scheds) == 1
    assert isinstance(opts[0], torch.optim.AdamW) and opts[0].defaults["lr"] == 4.0e-4 and opts[0].defaults["weight_decay"] == 4.0e-8
    assert isinstance(scheds[0], torch.optim.lr_scheduler.StepLR) and scheds[0].step_size == 15 and scheds[0].gamma == 0.1
    assert sum(p.numel() for g in opts[0].param_groups for p in g["params"]) == sum(p.numel() for p in model.parameters())
    # hydra-style nodes (conf/optim/sgd.yaml-like, conf/scheduler/step.yaml)
    cfg = default_config(hidden_nf=32, n_layers=1)
    cfg["optim"] = {"_target_": "torch.optim.SGD", "lr": 0.1, "momentum": 0.9}
    cfg["scheduler"] = {"_target_": "torch.optim.lr_scheduler.StepLR", "step_size": 3, "gamma": 0.5}
    opts, scheds = DiffusionQM9(cfg).configure_optimizers()
    assert isinstance(opts[0], torch.optim.SGD) and opts[0].defaults["momentum"] == 0.9 and scheds[0].step_size == 3
    cfg["optim"] = {"_target_": "os.system", "command": "true"}
    with pytest.raises(ValueError):
        DiffusionQM9(cfg).configure_optimizers()
    # two ranks: the gather joins the ranks' steps, both ranks see the same mean, only rank 0 logs the test metric
    world, port = 2, _free_port()
    mp.spawn(_epoch_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"e{k}.pt") for k in range(world)]
    want = np.mean([1.0, 11.0, 21.0, 2.0, 12.0, 22.0])
    assert r[0]["logged"]["val_loss"] == r[1]["logged"]["val_loss"] == pytest.approx(want)
    assert r[0]["logged"]["test/ppl"] == pytest.approx(want) and "test/ppl" not in r[1]["logged"]
    assert torch.equal(r[0]["gathered"], torch.tensor([0.0, 5.0, 1.0, 5.0])) and torch.equal(r[0]["gathered"], r[1]["gathered"])


def test_epoch_end_hooks_use_the_lightning_base_when_it_exists(tmp_path):
    """With pytorch_lightning installed `DiffusionQM9` derives from LightningModule (hierdiff_amd/diffusion.py: `_Base`) and its hooks
    must go through the base's `log` / `all_gather` / `global_rank`, like the reference's (diffusion_qm9.py:753-801).  The image has no
    Lightning: a STUB package with that surface is put in front of the import in a fresh interpreter (the same two-line stub the
    oracle generator uses to import the reference)."""
    import subprocess
    import sys
    import textwrap
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / "pytorch_lightning"
    stub.mkdir()
    (stub / "__init__.py").write_text(textwrap.dedent('''
        import torch
        class LightningModule(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.logged_by_base, self.gathers, self.global_rank = {}, 0, 1
            def log(self, name, value, **kw):
                self.logged_by_base[name] = (float(value), kw)
            def all_gather(self, t):
                self.gathers += 1
                return torch.stack([t, t + 100.0])          # a world of two: this rank and a peer whose values are 100 larger
            def save_hyperparameters(self, *a, **k):
                pass
    '''))
    code = textwrap.dedent('''
        import sys, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import pytorch_lightning as pl
        from hierdiff_amd import DiffusionQM9, default_config
        m = DiffusionQM9(default_config(hidden_nf=32, n_layers=1))
        assert isinstance(m, pl.LightningModule)
        steps = [{"loss": torch.tensor(1.0)}, {"loss": torch.tensor(3.0)}]
        m.validation_epoch_end(steps)
        assert m.gathers == 1 and abs(m.logged_by_base["val_loss"][0] - 52.0) < 1e-6, m.logged_by_base     # mean(1, 3, 101, 103)
        assert m.logged_by_base["val_loss"][1] == {"on_epoch": True, "prog_bar": True}
        m.test_epoch_end(steps)
        assert "test/ppl" not in m.logged_by_base          # global_rank 1: only rank 0 logs the test metric
        m.global_rank = 0
        m.test_epoch_end(steps)
        assert abs(m.logged_by_base["test/ppl"][0] - 52.0) < 1e-6 and not hasattr(m, "logged")
        opts, scheds = m.configure_optimizers()
        assert len(opts) == len(scheds) == 1
        print("LIGHTNING-HOOKS-OK")
    ''') % (str(tmp_path), repo)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "LIGHTNING-HOOKS-OK" in out.stdout, out.stderr[-3000:]
