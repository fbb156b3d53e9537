"""Drop-in for the reference's `EGNN_dynamics_QM9` (endiffusion/models/module/en_dynamics.py:8-143).

Same constructor, same `state_dict` keys (`egnn.embedding.weight`, `egnn.e_block_{i}.gcl_{j}.edge_mlp.0.weight`,
... -- SURVEY.md section 8b), same `_forward(t, xh, node_mask, edge_mask, context, mol_shape=None)`
signature and return value, so it nests inside the reference's `DiffusionQM9` / Lightning pipeline
unchanged.  The arithmetic runs in hand-written HIP kernels behind the C ABI of
include/hierdiff_hip.h; the torch modules below only hold parameters.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import math
import os
import weakref
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import HdConfig, HierDiffHipError


PRECISIONS = {"fp32": 0, "fp16x3": 3}
RETIRED_PRECISIONS = ("bf16x3", "bf16x6")
# Arithmetic of the H x H contractions.  The drop-in default is "fp32": exact fp32 matrix instructions
# (v_mfma_f32_32x32x2_f32), the arithmetic the reference computes in (en_dynamics.py has no notion of reduced
# precision; per-forward error vs the reference ~5e-7 rel-L2).  "fp16x3" is the one opt-in (`model.precision = "fp16x3"` or
# HIERDIFF_PRECISION=fp16x3): fp32-ACCURATE on the matrix cores proper - a two-way FP16 split with operands ranged by exact powers
# of two, three MFMAs per product, as far from a float64 evaluation as exact fp32, 2.4x faster end to end; the mode to use for
# sampling.  The bf16 splits of rounds 1-5 were retired in round 6: "bf16x6" (same accuracy, 1.6x slower than fp16x3) and
# "bf16x3" (4 % faster at 20x the error) were dominated on both axes.
DEFAULT_PRECISION = "fp32"

def _checked_precision(name: str) -> str:
    if name in RETIRED_PRECISIONS:
        raise ValueError(f'precision {name!r} was retired in round 6 (ABI 12): use "fp16x3" - as accurate as "bf16x6" and exact fp32, '
                         f'at the speed of "bf16x3" - or "fp32"')
    if name not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
    return name


# ----------------------------------------------------------------------------- parameter holders
# Mirrors of the reference module tree; they are never called, only hold tensors so that
# state_dict()/load_state_dict()/.to()/DDP-style broadcast see the reference's key layout.

class _GCLParams(nn.Module):
    """egnn_new.py:9-33"""

    def __init__(self, hidden_nf: int, edges_in_d: int, attention: bool):
        super().__init__()
        self.edge_mlp = nn.Sequential(nn.Linear(2 * hidden_nf + edges_in_d, hidden_nf), nn.SiLU(),
                                      nn.Linear(hidden_nf, hidden_nf), nn.SiLU())
        self.node_mlp = nn.Sequential(nn.Linear(2 * hidden_nf, hidden_nf), nn.SiLU(),
                                      nn.Linear(hidden_nf, hidden_nf))
        if attention:
            self.att_mlp = nn.Sequential(nn.Linear(hidden_nf, 1), nn.Sigmoid())


class _EquivariantUpdateParams(nn.Module):
    """egnn_new.py:74-89"""

    def __init__(self, hidden_nf: int, edges_in_d: int):
        super().__init__()
        layer = nn.Linear(hidden_nf, 1, bias=False)
        torch.nn.init.xavier_uniform_(layer.weight, gain=0.001)
        self.coord_mlp = nn.Sequential(nn.Linear(2 * hidden_nf + edges_in_d, hidden_nf), nn.SiLU(),
                                       nn.Linear(hidden_nf, hidden_nf), nn.SiLU(), layer)


class _EquivariantBlockParams(nn.Module):
    """egnn_new.py:114-137"""

    def __init__(self, hidden_nf: int, n_layers: int, attention: bool):
        super().__init__()
        for i in range(n_layers):
            self.add_module("gcl_%d" % i, _GCLParams(hidden_nf, 2, attention))
        self.add_module("gcl_equiv", _EquivariantUpdateParams(hidden_nf, 2))


class _GNNParams(nn.Module):
    """egnn_new.py:208-231 (mode 'gnn_dynamics'): GCLs without edge attributes, outputs [velocity | h]."""

    def __init__(self, in_node_nf: int, hidden_nf: int, out_node_nf: int, n_layers: int, attention: bool):
        super().__init__()
        self.embedding = nn.Linear(in_node_nf, hidden_nf)
        self.embedding_out = nn.Linear(hidden_nf, out_node_nf)
        for i in range(n_layers):
            self.add_module("gcl_%d" % i, _GCLParams(hidden_nf, 0, attention))

    def forward(self, *a, **k):
        raise NotImplementedError("parameters only; the arithmetic lives in libhierdiff_hip.so")


class _EGNNParams(nn.Module):
    """egnn_new.py:155-190"""

    def __init__(self, in_node_nf: int, hidden_nf: int, n_layers: int, inv_sublayers: int, attention: bool):
        super().__init__()
        self.embedding = nn.Linear(in_node_nf, hidden_nf)
        self.embedding_out = nn.Linear(hidden_nf, in_node_nf)
        for i in range(n_layers):
            self.add_module("e_block_%d" % i, _EquivariantBlockParams(hidden_nf, inv_sublayers, attention))

    def forward(self, *a, **k):
        raise NotImplementedError("parameters only; the arithmetic lives in libhierdiff_hip.so")


# ----------------------------------------------------------------------------- topology cache

class Topology:
    """Index tables + activation workspace for one (node_mask, edge_mask) pair (hd_topology)."""

    def __init__(self, owner: "EGNN_dynamics_QM9", node_mask: torch.Tensor, edge_mask: Optional[torch.Tensor],
                 B: int, N: int, host_masks: Optional[Tuple[np.ndarray, Optional[np.ndarray]]] = None,
                 device: Optional[torch.device] = None):
        lib = _lib.load()
        nm, em = host_masks if host_masks is not None else masks_to_host(node_mask, edge_mask, B, N)
        self._h = C.c_void_p()
        # tables upload in stream order of the current stream, no host wait (hd_topology_create_s)
        dev = owner._device() if device is None else device
        with torch.cuda.device(dev):
            _lib.check(lib.hd_topology_create_s(owner._handle(), nm.ctypes.data, None if em is None else em.ctypes.data, B, N,
                                                _stream(dev), C.byref(self._h)), "hd_topology_create_s")
        self.B, self.N = B, N
        self._finalizer = weakref.finalize(self, lib.hd_topology_destroy, self._h)

    @property
    def ptr(self) -> C.c_void_p:
        return self._h

    def info(self) -> Dict[str, int]:
        buf = (C.c_longlong * 6)()
        _lib.check(_lib.load().hd_topology_info(self._h, buf), "hd_topology_info")
        return dict(zip(("B", "N", "nodes", "edges", "tiles", "parts"), (int(v) for v in buf)))


def masks_to_host(node_mask: torch.Tensor, edge_mask: Optional[torch.Tensor], B: int, N: int):
    """(node mask [B*N], edge mask [B*N*N] or None) as host uint8 arrays with ONE device-to-host copy (one sync)."""
    nm = node_mask.reshape(B * N).to(torch.bool)
    if edge_mask is None:
        return nm.cpu().numpy().astype(np.uint8), None
    both = torch.cat([nm.view(torch.uint8), edge_mask.reshape(B * N * N).to(torch.bool).view(torch.uint8)]).cpu().numpy()
    return np.ascontiguousarray(both[:B * N]), np.ascontiguousarray(both[B * N:])


def release_cached_memory() -> None:
    """Frees the device arenas (and their pinned twins) the library keeps for recycled topologies - memory torch's caching
    allocator cannot see.  The counterpart of torch.cuda.empty_cache(): call it before a phase that needs the room; the pool
    refills on demand.  (hd_arena_pool_trim; also called when a dynamics module releases its handle.)"""
    try:
        _lib.load().hd_arena_pool_trim()
    except Exception:
        pass


_PIN_RING: Dict[Tuple, list] = {}
_PIN_DEPTH = 4                      # slots a byte class starts with
_PIN_CLASS_BYTES = 64 << 20         # ... and may grow to (per class and device) before a staging call waits for a copy


def _to_device_async(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    """Host tensor -> device in stream order WITHOUT a host wait: through a pinned staging buffer this module keeps (a ring of
    at least _PIN_DEPTH buffers per byte class, each guarded by the event behind its last copy; a buffer is reused only once that
    event has completed, the ring grows instead - up to _PIN_CLASS_BYTES - when every buffer is still in flight).  What it avoids, all measured in a
    staged training loop at B = 256 (scratch/fresh_masks_trace.py, scratch/busy_gpu_experiments.py): a pageable source makes torch
    wait for the stream (the host cannot run ahead); `Tensor.pin_memory()` allocates, and with the host ahead of the GPU its cached
    blocks are still in flight, so it falls through to hipHostMalloc, which waits for the device (60 ms stalls); and
    `pinned.copy_(t)` fans tensors of 32 K elements and more out over torch's intra-op thread pool - on a 128-core host those
    spinning workers starve the runtime's completion-signal thread: 50 - 90 ms stalls of every other training step when a 230 KB
    mask was staged that way.  Hence one plain memcpy into the pinned slot."""
    if t.device.type != "cpu":
        return t.to(device)
    dev = torch.device(device)
    nbytes = t.numel() * t.element_size()
    # rings are keyed by a rounded-up BYTE size, not by shape: a loader that pads every batch to its own n_max yields dozens of
    # shapes per epoch, and a ring per shape would keep evicting and re-allocating pinned memory (hipHostMalloc waits for the
    # device - the stall this function exists to avoid).  Powers of two from 4 KiB: at most ~20 rings per device.
    cap = 4096
    while cap < nbytes:
        cap *= 2
    key = (cap, str(dev))
    ring = _PIN_RING.get(key)
    if ring is None:
        ring = _PIN_RING[key] = [0, []]
    pos, slots = ring
    slot = None
    if len(slots) >= _PIN_DEPTH:
        # oldest slot first; a slot whose copy has not run yet (tensors of ONE batch share a byte class, and their copies queue
        # behind the training step that is still executing) must not be waited for - that wait is the stall this function exists
        # to avoid: take the next finished slot, or grow the ring while the class stays under its byte cap
        for k in range(len(slots)):
            cand = slots[(pos + k) % len(slots)]
            if cand[1] is None or cand[1].query():
                slot = cand
                ring[0] = (pos + k + 1) % len(slots)
                break
        if slot is None and len(slots) * cap >= _PIN_CLASS_BYTES:
            slot = slots[pos % len(slots)]
            ring[0] = (pos + 1) % len(slots)
            slot[1].synchronize()                                   # the cap is reached: wait for the oldest copy
    if slot is None:
        slots.append([torch.empty(cap, dtype=torch.uint8, pin_memory=True), None])
        slot = slots[-1]
    src = t.contiguous()
    if nbytes:
        C.memmove(slot[0].data_ptr(), src.data_ptr(), nbytes)
    staged = slot[0][:nbytes].view(t.dtype).view(t.shape)
    with torch.cuda.device(dev):
        out = staged.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
    slot[1] = ev
    return out


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _as_f32(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class EGNN_dynamics_QM9(nn.Module):
    """HIP implementation of en_dynamics.py:8-143 (mode 'egnn_dynamics')."""

    def __init__(self, in_node_nf, context_node_nf, n_dims, hidden_nf=64, act_fn="silu", n_layers=4,
                 attention=False, condition_time=True, tanh=False, mode='egnn_dynamics', norm_constant=0,
                 inv_sublayers=2, sin_embedding=False, normalization_factor=100, aggregation_method='sum'):
        super().__init__()
        if mode not in ('egnn_dynamics', 'gnn_dynamics'):
            raise Exception("Wrong mode %s" % mode)                  # en_dynamics.py:96-97
        if sin_embedding:
            raise NotImplementedError("sin_embedding=True is config-off in the reference (ddpmgblur.yaml:35)")
        if aggregation_method not in ('sum', 'mean'):
            raise ValueError(f"aggregation_method {aggregation_method!r}: 'sum' or 'mean' (egnn_new.py:269-289)")
        if not (act_fn == "silu" or isinstance(act_fn, nn.SiLU)):
            raise NotImplementedError("act_fn must be 'silu'")
        if n_dims != 3:
            raise NotImplementedError("n_dims must be 3")
        if hidden_nf not in (32, 64, 128, 256):
            raise NotImplementedError("hidden_nf must be one of 32, 64, 128, 256")
        self.mode = mode
        if mode == 'gnn_dynamics':
            self._init_gnn(in_node_nf, context_node_nf, n_dims, hidden_nf, n_layers, attention, condition_time,
                           normalization_factor, aggregation_method)
            return
        self.egnn = _EGNNParams(in_node_nf + context_node_nf, hidden_nf, n_layers, inv_sublayers, bool(attention))
        self.in_node_nf = in_node_nf
        self.context_node_nf = context_node_nf
        self.n_dims = n_dims
        self.condition_time = condition_time
        self._edges_dict = {}       # kept for attribute parity with the reference; unused
        self._cfg = HdConfig(in_node_nf=in_node_nf, context_node_nf=context_node_nf, n_dims=n_dims,
                             hidden_nf=hidden_nf, n_layers=n_layers, inv_sublayers=inv_sublayers,
                             attention=int(bool(attention)), tanh=int(bool(tanh)),
                             condition_time=int(bool(condition_time)), norm_constant=float(norm_constant),
                             normalization_factor=float(normalization_factor), coords_range=30.0,
                             precision=PRECISIONS[_checked_precision(os.environ.get("HIERDIFF_PRECISION", DEFAULT_PRECISION))],
                             aggregation_mean=int(aggregation_method == 'mean'))
        self.aggregation_method = aggregation_method
        self._hd = None              # (handle, device index)
        self._handle_gen = 0         # bumped whenever a new hd_handle is created (schedule / weights must be re-sent)
        self._weights_key = None
        self._topo_cache: Dict[Tuple, Tuple] = {}               # by mask tensor identity
        self._topo_by_content: Dict[Tuple, Topology] = {}       # by mask content
        self.debug_checks = False
        self.differentiable: Optional[bool] = None      # None: by rule (_wants_autograd)

    # ------------------------------------------------------------------ mode 'gnn_dynamics' (en_dynamics.py:24-29, 91-94)
    # The non-equivariant variant: GNN (egnn_new.py:208-242) over the node inputs [x | h | t], GCLs without edge attributes, the
    # first three output columns are the velocity.  The reference calls it WITHOUT an edge mask (en_dynamics.py:93), so the
    # messages run over all N x N pairs of a molecule - self pairs and padded nodes included - and only the rows are masked.
    # It runs on the same kernels: an internal egnn-mode engine with ONE block of n_layers GCLs, the coordinates fed as features
    # (its own coordinates are zero, so the two distance columns it adds to the first edge Linear - zero weights - and its
    # coordinate layer - zero weights - contribute exact zeros), an all-ones edge mask (which makes every node of the batch an
    # active node of the layout while the row mask stays the node mask), and the engine's output columns read as
    # [unused | velocity | h].  Inference only; DiffusionQM9 samples with it through its step-by-step loop (the library's fused
    # sampler loop and the training path are egnn_dynamics').
    def _init_gnn(self, in_node_nf, context_node_nf, n_dims, hidden_nf, n_layers, attention, condition_time, normalization_factor,
                  aggregation_method):
        if context_node_nf:
            raise NotImplementedError("mode 'gnn_dynamics' with context: the reference strips context columns its GNN never "
                                      "produced (en_dynamics.py:28, 99-101) and returns a tensor of the wrong width")
        self.gnn = _GNNParams(in_node_nf + context_node_nf + 3, hidden_nf, 3 + in_node_nf, n_layers, bool(attention))
        self.in_node_nf, self.context_node_nf, self.n_dims, self.condition_time = in_node_nf, context_node_nf, n_dims, condition_time
        self.aggregation_method = aggregation_method
        self._edges_dict = {}
        self.debug_checks = False
        self.differentiable = None
        self._hd, self._handle_gen, self._weights_key, self._topo_cache, self._topo_by_content = None, 0, None, {}, {}
        engine = EGNN_dynamics_QM9(in_node_nf + 3, 0, n_dims, hidden_nf=hidden_nf, n_layers=1, attention=attention,
                                   condition_time=condition_time, tanh=False, norm_constant=1, inv_sublayers=n_layers,
                                   normalization_factor=normalization_factor, aggregation_method=aggregation_method)
        for p in engine.parameters():
            p.requires_grad_(False)
        object.__setattr__(self, "_engine", engine)          # not a submodule: its tensors are derived, not part of the state_dict
        self._engine_key = None
        # the sampler's own arithmetic (noise, posterior step, decode: hd_noise / hd_posterior_step / hd_final_decode) needs a handle
        # and tile tables of THIS module's [B, N, 3 + F] layout, not of the engine's widened one: a minimal egnn-mode twin provides
        # them (its network is never evaluated)
        arith = EGNN_dynamics_QM9(in_node_nf, 0, n_dims, hidden_nf=32, n_layers=1, attention=False, condition_time=condition_time,
                                  inv_sublayers=1, normalization_factor=1)
        for p in arith.parameters():
            p.requires_grad_(False)
        object.__setattr__(self, "_arith", arith)

    def _sync_gnn_engine(self):
        params = list(self.gnn.parameters())
        key = (_lib.optimizer_generation(),) + tuple((p.data_ptr(), p._version) for p in params)
        guard = self.__dict__.setdefault("_engine_guard", _lib.ImageGuard())
        if self._engine_key is None:
            guard.clear()
        if guard.valid(key, params):
            return
        eg, gn = self._engine.egnn, self.gnn
        H = gn.embedding.weight.shape[0]
        with torch.no_grad():
            eg.embedding.weight.copy_(gn.embedding.weight); eg.embedding.bias.copy_(gn.embedding.bias)
            eg.embedding_out.weight.copy_(gn.embedding_out.weight); eg.embedding_out.bias.copy_(gn.embedding_out.bias)
            blk = eg.e_block_0
            for i in range(len([k for k in gn._modules if k.startswith("gcl_")])):
                src, dst = getattr(gn, f"gcl_{i}"), getattr(blk, f"gcl_{i}")
                dst.edge_mlp[0].weight.zero_()
                dst.edge_mlp[0].weight[:, :2 * H].copy_(src.edge_mlp[0].weight)     # the two distance columns stay zero
                dst.edge_mlp[0].bias.copy_(src.edge_mlp[0].bias)
                for a, b in ((dst.edge_mlp[2], src.edge_mlp[2]), (dst.node_mlp[0], src.node_mlp[0]), (dst.node_mlp[2], src.node_mlp[2])):
                    a.weight.copy_(b.weight); a.bias.copy_(b.bias)
                if hasattr(src, "att_mlp"):
                    dst.att_mlp[0].weight.copy_(src.att_mlp[0].weight); dst.att_mlp[0].bias.copy_(src.att_mlp[0].bias)
            for p in blk.gcl_equiv.parameters():
                p.zero_()
        guard.store(key, params)
        self._engine_key = key

    def _forward_gnn(self, t, xh, node_mask, edge_mask, context):
        if torch.is_grad_enabled() and (xh.requires_grad or (self.training and any(p.requires_grad for p in self.gnn.parameters()))) \
                and self.differentiable is not False:
            raise NotImplementedError("mode 'gnn_dynamics' is inference only here: call it under torch.no_grad() / model.eval()")
        bs, n_nodes, dims = xh.shape
        if dims - self.n_dims != self.in_node_nf - (1 if self.condition_time else 0):
            raise ValueError(f"xh has {dims - self.n_dims} feature columns, model expects "
                             f"{self.in_node_nf - (1 if self.condition_time else 0)}")
        self._sync_gnn_engine()
        nm = node_mask.reshape(bs, n_nodes, 1)
        with torch.no_grad():
            xh_m = xh.detach().to(torch.float32) * nm.to(torch.float32)
            feed = torch.cat([torch.zeros_like(xh_m[:, :, :self.n_dims]), xh_m], dim=2)        # [coordinates = 0 | x | h]
            all_pairs = torch.ones((bs, n_nodes, n_nodes), dtype=torch.bool, device=xh.device)
            out = self._engine._forward(t, feed, nm, all_pairs, None, None)
            vel, h_final = out[:, :, self.n_dims:2 * self.n_dims], out[:, :, 2 * self.n_dims:]
            vel = torch.where(torch.isnan(vel).any(), torch.zeros_like(vel), vel)              # en_dynamics.py:109-111, no host sync
            nmf = nm.to(torch.float32)
            vel = vel - (vel.sum(1, keepdim=True) / nmf.sum(1, keepdim=True)) * nmf            # remove_mean_with_mask (:116)
            return torch.cat([vel, h_final], dim=2)

    # ------------------------------------------------------------------ arithmetic of the TRAINING path
    @property
    def training_precision(self) -> str:
        """"fp32" (default): every kernel of a training step is exact fp32.  "fp16x3" (opt-in; hidden_nf >= 128): the forward's
        per-edge H x H contraction, stage B's dP = G2 W2 and the dense reduction dW2 = G2^T P run on the matrix cores proper in the
        sampler's fp32-ACCURATE two-way FP16 split (three MFMAs per product; operand rows / arrays ranged by exact powers of two
        computed on the device from the data itself) on top of the kept pre-activations (`keep_edge_activations`), while everything
        around them - first-layer recomputation, SiLU and its derivative, the node-level GEMMs and the loss - stays exact fp32:
        this implementation's counterpart of the reference's mixed-precision training (apex O2, conf/trainer/default.yaml:4-5)
        without its loss of accuracy (gradients agree with the exact-fp32 step to ~1e-6, tests/test_gpu_training.py).  A layer
        whose batch is too small to keep its pre-activations runs in exact fp32 (a one-time warning says so)."""
        return getattr(self, "_training_precision", "fp32")

    @training_precision.setter
    def training_precision(self, name: str) -> None:
        if name == "bf16x6":
            raise ValueError('training_precision "bf16x6" was retired in round 6 (ABI 12): "fp16x3" is as accurate and faster')
        if name not in ("fp32", "fp16x3"):
            raise ValueError('training_precision must be "fp32" or "fp16x3"')
        object.__setattr__(self, "_training_precision", name)

    #: Training forward keeps the second-layer pre-activations W2 P + b2 of every edge row for its backward pass where the
    #: whole-tile edge kernel runs (large batches; `hd_edge_layer_save_rows`), so stage A of the backward pass loads them instead
    #: of recomputing them on the matrix cores: [edge rows, hidden_nf] fp32 per edge layer - 228 MB x 18 layers = 4.1 GB at
    #: B = 256, N = 30, H = 256, a size chosen for this GPU's 288 GB; it grows as B N^2 H layers, and a layer whose buffer cannot be
    #: allocated falls back to recomputing (one warning).  False = recompute everywhere (rounds 2-4; same gradients to the bit).
    keep_edge_activations = True

    # ------------------------------------------------------------------ precision of the matrix-core path
    @property
    def precision(self) -> str:
        if self.mode == 'gnn_dynamics':
            return self._engine.precision
        return {v: k for k, v in PRECISIONS.items()}[self._cfg.precision]

    @precision.setter
    def precision(self, name: str) -> None:
        """"fp32": exact fp32 matrix instructions.  "fp16x3": edge and node contractions on a two-way FP16 split (three fp16 matrix
        instructions per product), every operand ranged by an exact power of two per matrix / per row - as accurate as "fp32" at
        2.4x its speed; the recommended sampling mode."""
        _checked_precision(name)
        if self.mode == 'gnn_dynamics':
            self._engine.precision = name
            return
        if PRECISIONS[name] != self._cfg.precision:
            self._release()
            self._cfg.precision = PRECISIONS[name]

    # ------------------------------------------------------------------ reference API surface
    def forward(self, t, xh, node_mask, edge_mask, context=None):
        raise NotImplementedError

    def wrap_forward(self, node_mask, edge_mask, context):
        def fwd(time, state):
            return self._forward(time, state, node_mask, edge_mask, context)
        return fwd

    def unwrap_forward(self):
        return self._forward

    # ------------------------------------------------------------------ handle / weights
    def _device(self) -> torch.device:
        return (self.gnn if self.mode == 'gnn_dynamics' else self.egnn).embedding.weight.device

    def _handle(self) -> C.c_void_p:
        if self.mode == 'gnn_dynamics':
            return self._arith._handle()             # sampler arithmetic only; the network runs through `_forward`
        dev = self._device()
        if dev.type != "cuda":
            raise HierDiffHipError("EGNN_dynamics_QM9 runs only on an MI355X: move the module to a cuda device "
                                   "(there is no CPU fallback)")
        _lib.require_gpu()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._hd is None or self._hd[1] != idx:
            self._release()
            lib = _lib.load()
            h = C.c_void_p()
            _lib.check(lib.hd_create(C.byref(self._cfg), idx, C.byref(h)), "hd_create")
            self._hd = (h, idx)
            self._handle_gen += 1
            self._weights_key = None
            self._finalizer = weakref.finalize(self, lib.hd_destroy, h)
        return self._hd[0]

    def _release(self):
        self._topo_cache.clear()
        self._topo_by_content.clear()
        if self._hd is not None:
            self._finalizer()
            self._hd = None
            release_cached_memory()      # the topologies' arenas went back to the library's pool: hand them to the driver

    def canonical_blob(self) -> torch.Tensor:
        """Parameters flattened in registration order (== hierdiff_amd.weights.flatten_dynamics)."""
        return torch.cat([p.detach().reshape(-1).to(torch.float32) for p in self.egnn.parameters()])

    def sync_weights(self, force: bool = False) -> None:
        """(Re)pack the parameters into the HIP handle when they changed since the last call."""
        if self.mode == 'gnn_dynamics':
            self._sync_gnn_engine()
            self._arith.sync_weights(force)
            self._handle_gen = self._arith._handle_gen
            return
        h = self._handle()
        # cheap key: (address, in-place version) of every parameter + the optimizer-step count (fused optimizers do not bump
        # `_version`, _lib.optimizer_generation).  A key hit is then CONFIRMED by content (_lib.ImageGuard: one digest launch
        # over the parameters, 8 bytes read back), so a writer that bumps nothing - `p.data.copy_`, an external kernel - cannot
        # leave the handle on a stale image either.
        params = list(self.egnn.parameters())
        key = (_lib.optimizer_generation(),) + tuple((p.data_ptr(), p._version) for p in params)
        guard = self.__dict__.setdefault("_weights_guard", _lib.ImageGuard())
        if self._weights_key is None:
            guard.clear()
        if not force and guard.valid(key, params):
            return
        blob = self.canonical_blob().contiguous()
        lib = _lib.load()
        expect = lib.hd_weight_count(h)
        if blob.numel() != expect:
            raise HierDiffHipError(f"parameter count {blob.numel()} != library layout {expect}")
        _lib.check(lib.hd_set_weights(h, blob.data_ptr(), blob.numel(), 1, _stream(blob.device)), "hd_set_weights")
        guard.store(key, params)
        self._weights_key = key

    # ------------------------------------------------------------------ topology
    def topology(self, node_mask: torch.Tensor, edge_mask: Optional[torch.Tensor], B: int, N: int) -> Topology:
        """Index tables + workspace for a pair of masks, cached twice over.  Fast path: the mask TENSORS of an earlier call
        (storage address + in-place version + shape: the 1001 forwards of a sampling run, and views of one mask tensor - the
        `.view(bs, n*n)` the reference's forward(batch) makes every step - hit without touching the data; like the reference
        caches its edge lists per (n_nodes, batch_size) in `_edges_dict`, en_dynamics.py:124-143).  Otherwise - a training
        loop hands over NEW mask tensors every batch - the key is the masks' CONTENT (one device-to-host copy, which
        building the tables needs anyway, and a 16-byte digest): batches whose masks repeat share one topology, and a
        miss costs one allocation, one upload and one memset (hd_topology_create)."""
        if self.mode == 'gnn_dynamics':
            return self._arith.topology(node_mask, edge_mask, B, N)
        self._handle()
        sig = lambda m: None if m is None else (m.data_ptr(), m._version, m.numel(), m.dtype, str(m.device))
        key = (sig(node_mask), sig(edge_mask), B, N)
        hit = self._topo_cache.get(key)
        if hit is not None:
            return hit[0]
        nm, em = masks_to_host(node_mask, edge_mask, B, N)
        topo = self._topology_by_content(nm, em, B, N)
        self._remember(key, topo, node_mask, edge_mask)
        return topo

    def _topology_by_content(self, nm: np.ndarray, em: Optional[np.ndarray], B: int, N: int) -> Topology:
        dig = hashlib.blake2b(nm.tobytes() + (b"" if em is None else em.tobytes()), digest_size=16).digest()
        ckey = (dig, em is None, B, N)
        topo = self._topo_by_content.get(ckey)
        if topo is None:
            if len(self._topo_by_content) >= 8:
                self._topo_by_content.pop(next(iter(self._topo_by_content)))
            topo = Topology(self, None, None, B, N, host_masks=(nm, em))
            self._topo_by_content[ckey] = topo
        else:
            self._topo_by_content[ckey] = self._topo_by_content.pop(ckey)      # most recently used last
        return topo

    def _remember(self, key, topo, node_mask, edge_mask) -> None:
        if len(self._topo_cache) >= 8:
            self._topo_cache.pop(next(iter(self._topo_cache)))
        self._topo_cache[key] = (topo, node_mask, edge_mask)     # holding the tensors pins their addresses

    def stage_masks(self, node_mask: torch.Tensor, edge_mask: Optional[torch.Tensor], device: Optional[torch.device] = None):
        """Host masks in, device masks + their topology out, WITHOUT a host wait: what a training loop does with the
        masks a DataLoader hands it (the reference's batches are collated on the host and moved by Lightning's
        transfer_batch_to_device).  The index tables are laid out from the host tensors (no device-to-host copy, which would
        stall the loop behind everything the GPU still has queued), uploaded in stream order from pinned staging into a pooled
        arena (hd_topology_create_s), and the device copies of the masks are registered as this topology's identity -
        `_forward(t, xh, node_mask, edge_mask, ...)` on them (or on views of them) finds it without touching the data.
        Returns (node_mask_dev, edge_mask_dev)."""
        if node_mask.device.type != "cpu" or (edge_mask is not None and edge_mask.device.type != "cpu"):
            raise HierDiffHipError("stage_masks takes the HOST masks of a collated batch")
        dev = self._device() if device is None else torch.device(device)
        B = node_mask.shape[0]
        N = node_mask.numel() // B
        as_bytes = lambda m, n: np.ascontiguousarray(m.numpy().reshape(n) != 0).view(np.uint8)     # numpy: single-threaded
        nm = as_bytes(node_mask, B * N)
        em = None if edge_mask is None else as_bytes(edge_mask, B * N * N)
        if self.mode == 'gnn_dynamics':
            return node_mask.to(dev, non_blocking=True), None if edge_mask is None else edge_mask.to(dev, non_blocking=True)
        self._handle()
        topo = self._topology_by_content(nm, em, B, N)
        nm_d = _to_device_async(node_mask, dev)
        em_d = None if edge_mask is None else _to_device_async(edge_mask, dev)
        sig = lambda m: None if m is None else (m.data_ptr(), m._version, m.numel(), m.dtype, str(m.device))
        self._remember((sig(nm_d), sig(em_d), B, N), topo, nm_d, em_d)
        return nm_d, em_d

    # ------------------------------------------------------------------ forward
    def _forward(self, t, xh, node_mask, edge_mask, context, mol_shape=None):
        """en_dynamics.py:49-122.  Returns a new [B, N, 3+F] fp32 tensor on xh.device.

        Like the reference module it is differentiable when autograd is recording (training_step: the exact-fp32 edge
        kernels and their hand-written backward, hierdiff_amd/training.py); under torch.no_grad() - sampling, validation -
        it is the inference path (hd_egnn_forward)."""
        if xh.device.type != "cuda":
            raise HierDiffHipError("EGNN_dynamics_QM9._forward needs cuda tensors (no CPU fallback)")
        if self.mode == 'gnn_dynamics':
            return self._forward_gnn(t, xh, node_mask, edge_mask, context)
        if self._wants_autograd(xh):
            from .training import dynamics_forward_train
            return dynamics_forward_train(self, t, xh, node_mask, edge_mask, context, mol_shape)
        bs, n_nodes, dims = xh.shape
        if dims - self.n_dims != self.in_node_nf - (1 if self.condition_time else 0):
            raise ValueError(f"xh has {dims - self.n_dims} feature columns, model expects "
                             f"{self.in_node_nf - (1 if self.condition_time else 0)}")
        self.sync_weights()
        topo = self.topology(node_mask, edge_mask, bs, n_nodes)
        return self.forward_with_topology(topo, t, xh, context, mol_shape)

    def _wants_autograd(self, xh: torch.Tensor) -> bool:
        """The differentiable (exact-fp32, training.py) path is taken when autograd is recording AND someone can use the
        graph: the module is in training mode with trainable parameters, or the input itself requires grad.  An evaluation
        call (`model.eval()`; validation NLL, sampling) made without an explicit torch.no_grad() therefore still runs the
        inference kernels of the configured precision - like the reference, whose eval-mode forward is the same arithmetic
        with or without no_grad.  `self.differentiable = True / False` overrides the rule."""
        if self.differentiable is not None:
            return bool(self.differentiable) and torch.is_grad_enabled()
        if not torch.is_grad_enabled():
            return False
        return xh.requires_grad or (self.training and any(p.requires_grad for p in self.egnn.parameters()))

    def forward_with_topology(self, topo: Topology, t, xh, context, mol_shape=None) -> torch.Tensor:
        dev = xh.device
        bs, n_nodes, dims = xh.shape
        xh_c = _as_f32(xh, dev)
        t_c = None
        t_numel = 1
        if self.condition_time:
            t_c = _as_f32(t, dev).reshape(-1)
            t_numel = t_c.numel()
            if t_numel not in (1, bs):
                raise ValueError("t must have 1 or batch_size elements")
        ctx_c = None
        if self.context_node_nf > 0:
            if context is None:
                raise ValueError("model has context_node_nf > 0 but context is None")
            ctx_c = _as_f32(context, dev).reshape(bs * n_nodes, self.context_node_nf)
        out = torch.empty((bs, n_nodes, dims), device=dev, dtype=torch.float32)
        _lib.check(_lib.load().hd_egnn_forward(self._handle(), topo.ptr, xh_c.data_ptr(), _ptr(t_c), t_numel,
                                               _ptr(ctx_c), -1 if mol_shape is None else int(mol_shape),
                                               out.data_ptr(), _stream(dev)), "hd_egnn_forward")
        if self.debug_checks:
            cnt = C.c_longlong()
            _lib.check(_lib.load().hd_nan_events(self._handle(), _stream(dev), C.byref(cnt)))
            if cnt.value:
                print('Warning: detected nan, resetting EGNN output to zero.')
        return out

    def get_adj_matrix(self, n_nodes, batch_size):
        """Reference helper (en_dynamics.py:124-143); the HIP path never materialises the dense edge
        list, this exists for callers that want it."""
        ar = torch.arange(n_nodes)
        rows = ar.repeat_interleave(n_nodes).repeat(batch_size)
        cols = ar.repeat(n_nodes).repeat(batch_size)
        off = (torch.arange(batch_size) * n_nodes).repeat_interleave(n_nodes * n_nodes)
        return [rows + off, cols + off]

    def _apply(self, fn, *a, **k):
        # .to()/.cuda() replace parameter storage: force a re-pack on the next call
        self._weights_key = None
        if self.mode == 'gnn_dynamics':
            self._engine._apply(fn, *a, **k)
            self._arith._apply(fn, *a, **k)
            self._engine_key = None
        return super()._apply(fn, *a, **k)

    def load_numpy_state_dict(self, sd: Dict[str, np.ndarray], prefix: str = "") -> None:
        """Convenience: load a {name: ndarray} dict (e.g. hierdiff_amd.weights.synthetic_*)."""
        own = self.state_dict()
        with torch.no_grad():
            for k in own:
                own[k].copy_(torch.from_numpy(np.asarray(sd[prefix + k])).to(own[k].device))
        self._weights_key = None
