"""Training forward / backward of the EGNN dynamics on MI355X.

Reference: `DiffusionQM9.training_step` -> `forward(batch)` -> `nll` -> `compute_loss` -> `phi`
(endiffusion/train_module/diffusion_qm9.py:774-777, 701-751, 675-699, 530-673, 135-138) with PyTorch autograd through
`EGNN_dynamics_QM9._forward` (models/module/en_dynamics.py:49-122) and DDP gradient averaging
(conf/trainer/default.yaml:2-3).

Split of the work, by what is hot:
  * the EDGE model of every GCL / EquivariantUpdate (gather, factorised first Linear, SiLU, H x H Linear, SiLU, attention
    gate or coordinate head, neighbour sum - 85 % of the FLOPs) runs in hand-written HIP both ways:
    `hd_edge_layer_forward` (the sampler's k_edge) and `hd_edge_layer_backward` (k_edge_bwd: per-edge activations are
    recomputed tile by tile, never stored by the forward pass), wrapped as ONE autograd Function;
  * the node-level Linears around it (embedding, the two halves of the first edge Linear, node MLP, output layer) are
    dense GEMMs on [nodes, H] matrices and run on this library's own exact-fp32 GEMM (`hd_gemm_f32`, csrc/k_tgemm.hpp)
    in all three directions - Y = X W^T + b, dX = dY W, dW = dY^T X (split-K, deterministic, the bias gradient riding
    along) - wrapped as the autograd Functions `_Linear` and `_NodeMLP` (the node MLP with its SiLU, residual and mask
    fused into the GEMM epilogues); so is the one dense reduction over all edge rows, dW2 = G2^T P.  No BLAS-library
    kernel is left in a training step.
Arithmetic is exact fp32 (`precision = "fp32"`), like the reference's training forward (its apex O2 mode is a
trainer-level choice outside this path).  Multi-GPU: one process per GPU, gradients averaged with one flat all-reduce
(`hierdiff_amd.sharding.allreduce_gradients`, RCCL under backend "nccl").
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .dynamics import EGNN_dynamics_QM9, Topology, _stream


class _TrainTables:
    """Per-topology helpers of the training path: compact node order on the device and the backward workspaces."""

    def __init__(self, topo: Topology, H: int, device: torch.device):
        lib = _lib.load()
        info = topo.info()
        self.M = info["nodes"]
        self.N = topo.N
        n_wg = (info["tiles"] + 3) // 4
        self.tiles = max(1, 4 * n_wg)
        self.rows = 32 * self.tiles
        # flat index b*N + n of the compact node order, written on the device from the topology's own table (a host array
        # would reach the device through a pageable copy = a stream synchronisation per new topology)
        self.index = torch.empty(self.M, dtype=torch.int64, device=device)
        if self.M > 0:                  # a batch without a single unmasked node: nothing to index (an empty tensor has no address)
            with torch.cuda.device(device):
                _lib.check(lib.hd_topology_nodes_device(topo.ptr, self.index.data_ptr(), _stream(device)), "hd_topology_nodes_device")
        self.H = H
        self.device = device

    def workspace(self):
        """Backward workspaces, SHARED by every topology of a (device, width): three [edge rows, H] buffers are 0.7 GB at
        B = 256, N = 30, H = 256, and a training loop builds a new topology for every batch of masks.  Grow-only; the
        views handed out cover this topology's rows (the kernels write every row of every tile, and the dense reduction
        dW2 = G2^T P reads exactly those rows)."""
        # keyed by the launching stream too: two models (or threads) running backward on different streams of one device
        # must not share these buffers
        pool = _WS_POOL.setdefault((str(self.device), self.H, int(_stream(self.device) or 0)), {"rows": 0, "tiles": 0})
        if pool["rows"] < self.rows or pool["tiles"] < self.tiles:
            z = lambda *s: torch.empty(s, device=self.device, dtype=torch.float32)
            pool["rows"], pool["tiles"] = max(pool["rows"], self.rows), max(pool["tiles"], self.tiles)
            pool.update(G2=z(pool["rows"], self.H), P=z(pool["rows"], self.H), G1=z(pool["rows"], self.H), escal=z(pool["rows"], 8),
                        colpart=z(pool["tiles"], self.H), bapart=z(pool["tiles"]), b2part=z(pool["tiles"], self.H),
                        wrdpart=z(pool["tiles"], 2, self.H))
        return dict(G2=pool["G2"][:self.rows], P=pool["P"][:self.rows], G1=pool["G1"][:self.rows], escal=pool["escal"][:self.rows],
                    colpart=pool["colpart"][:self.tiles], bapart=pool["bapart"][:self.tiles], b2part=pool["b2part"][:self.tiles],
                    wrdpart=pool["wrdpart"][:self.tiles])


_WS_POOL: dict = {}


def _tables(dyn: EGNN_dynamics_QM9, topo: Topology, device) -> _TrainTables:
    tr = getattr(topo, "_train", None)
    if tr is None:
        tr = _TrainTables(topo, dyn._cfg.hidden_nf, device)
        topo._train = tr
    return tr


# ----------------------------------------------------------------------------- dense GEMMs (hd_gemm_f32)

_EPI_BIAS, _EPI_BIAS_SILU2, _EPI_RESID_MASK, _EPI_MUL_DSILU = 0, 1, 2, 3


def _rows(t: torch.Tensor) -> torch.Tensor:
    """2-D fp32 view with unit column stride (any row stride)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.stride(1) == 1 or t.shape[1] == 1 and t.stride(0) >= 1 else t.contiguous()


def _dev_index(dev: torch.device) -> int:
    return torch.cuda.current_device() if dev.index is None else int(dev.index)


_SPLITK_WS: dict = {}


def _splitk_workspace(dev: torch.device, n: int) -> torch.Tensor:
    """Grow-only split-K partial-sum buffer per (device, stream): consumed by the reduce kernel of the same call, on the
    same stream, before the next GEMM of that stream can write it."""
    key = (_dev_index(dev), int(_stream(dev) or 0))
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 2 * ws.numel() if ws is not None else n), device=dev, dtype=torch.float32)
        _SPLITK_WS[key] = ws
    return ws


def _gemm(M, N, K, A, sam, sak, B, sbk, sbn, C, bias=None, epi=_EPI_BIAS, aux=None, rmask=None, C2=None, split=1,
          colsum=None):
    dev = C.device
    ws = None
    if split > 1:
        ws = _splitk_workspace(dev, split * (M * N + M))
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(_lib.load().hd_gemm_f32(_dev_index(dev), M, N, K, A.data_ptr(), sam, sak, B.data_ptr(), sbk, sbn, C.data_ptr(),
                                       C.stride(0), p(bias), epi, p(aux), p(rmask), p(C2), split, p(ws), p(colsum), _stream(dev)),
               "hd_gemm_f32")
    return C


def _linear_fwd(x, W, b, epi=_EPI_BIAS, aux=None, rmask=None):
    """epi(x W^T + b): returns C (and SiLU(C) for _EPI_BIAS_SILU2)."""
    M, K = x.shape
    N = W.shape[0]
    C = torch.empty((M, N), device=x.device, dtype=torch.float32)
    C2 = torch.empty_like(C) if epi == _EPI_BIAS_SILU2 else None
    _gemm(M, N, K, x, x.stride(0), 1, W, 1, W.stride(0), C, bias=b, epi=epi, aux=aux, rmask=rmask, C2=C2)
    return (C, C2) if C2 is not None else C


def _linear_dx(g, W, epi=_EPI_BIAS, aux=None):
    """epi(g W): [M, K] for g [M, N], W [N, K]."""
    M, N = g.shape
    K = W.shape[1]
    C = torch.empty((M, K), device=g.device, dtype=torch.float32)
    return _gemm(M, K, N, g, g.stride(0), 1, W, W.stride(0), 1, C, epi=epi, aux=aux)


def _split_for(m_out, n_out, k):
    """Slabs of a dW = dY^T X GEMM: enough workgroups to fill the chip (the result is only 8 - 16 tiles), at least eight
    K chunks per slab, and a workspace (slabs x result) that stays a few times the operands' size."""
    tiles = ((m_out + 63) // 64) * ((n_out + 127) // 128)
    if k < 512:
        return 1        # a short reduction (node rows of a small batch): one launch, no split-K reduce behind it
    return int(max(2, min(64, (512 + tiles - 1) // tiles, k // 256)))


def _linear_dw(g, x, want_db, rows=None):
    """dW = g^T x [N, K] (and db = column sums of g) over the first `rows` rows, split-K in slab order."""
    rows = g.shape[0] if rows is None else rows
    N, K = g.shape[1], x.shape[1]
    dW = torch.empty((N, K), device=g.device, dtype=torch.float32)
    db = torch.empty((N,), device=g.device, dtype=torch.float32) if want_db else None
    _gemm(N, K, rows, g, 1, g.stride(0), x, x.stride(0), 1, dW, split=_split_for(N, K, rows), colsum=db)
    return dW, db


class _Linear(torch.autograd.Function):
    """y = x W^T + b on hd_gemm_f32 (torch.nn.functional.linear's contract for 2-D x)."""

    @staticmethod
    def forward(ctx, x, W, b):
        x, W = _rows(x.detach()), _rows(W.detach())
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        return _linear_fwd(x, W, None if b is None else b.detach().contiguous())

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        g = _rows(gy)
        dx = _linear_dx(g, W) if ctx.needs_input_grad[0] else None
        dW, db = (None, None)
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dW, db = _linear_dw(g, x, ctx.has_b)
        return dx, dW, db


class _NodeMLP(torch.autograd.Function):
    """h' = (h + W4 SiLU(W3 [h ; agg] + b3) + b4) mask (egnn_new.py:57-69): two GEMMs forward (SiLU, residual and mask in
    their epilogues), four backward (SiLU' in the epilogue of dT = g W4)."""

    @staticmethod
    def forward(ctx, h, agg, W3, b3, W4, b4, nm):
        X = torch.cat([h.detach(), agg.detach()], dim=1)
        W3, W4 = _rows(W3.detach()), _rows(W4.detach())
        pre, T = _linear_fwd(X, W3, b3.detach().contiguous(), _EPI_BIAS_SILU2)
        hn = torch.empty((h.shape[0], W4.shape[0]), device=h.device, dtype=torch.float32)
        hres = h.detach().contiguous()
        _gemm(h.shape[0], W4.shape[0], W4.shape[1], T, T.stride(0), 1, W4, 1, W4.stride(0), hn, bias=b4.detach().contiguous(),
              epi=_EPI_RESID_MASK, aux=hres, rmask=nm)
        ctx.save_for_backward(X, pre, T, W3, W4, nm)
        return hn

    @staticmethod
    def backward(ctx, ghn):
        X, pre, T, W3, W4, nm = ctx.saved_tensors
        H = W4.shape[0]
        g = ghn * nm.unsqueeze(1)
        dW4, db4 = _linear_dw(g, T, True)
        dpre = _linear_dx(g, W4, _EPI_MUL_DSILU, aux=pre)       # (g W4) * SiLU'(pre)
        dW3, db3 = _linear_dw(dpre, X, True)
        dX = _linear_dx(dpre, W3)
        return dX[:, :H] + g, dX[:, H:], dW3, db3, dW4, db4, None


_PREC_CODE = {"fp32": 0, "fp16x3": 3}          # `precision` of hd_edge_layer_forward_s / _backward_s
_WARNED = set()
_F16WS = {}


def _f16_workspace(dev: torch.device, n: int) -> torch.Tensor:
    """Workspace of the fp16x3 backward (image scalars, maxima): one grow-only buffer per device AND stream, reused by every edge
    layer - the backward calls of a step are stream-ordered and each reads only what its own stages wrote (two models training on two
    streams of one device get two buffers)."""
    key = (dev.type, dev.index, int(_stream(dev) or 0))
    buf = _F16WS.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1024), device=dev, dtype=torch.float32)
        _F16WS[key] = buf
    return buf


class _EdgeLayer(torch.autograd.Function):
    """out = neighbour sum of the edge model of one GCL ([M, H]) or EquivariantUpdate ([M, 4], xyz0), from the node features:
    the factorised first edge Linear [A | B] = h [W1a ; W1b]^T + [b1 | 0] (one GEMM over the stacked halves of `W1`
    [H, 2H + 2]; its last two columns are the distance weights w_r, w_d) and the edge kernels of include/hierdiff_hip.h
    `hd_edge_layer_forward` / `hd_edge_layer_backward`.  One Function for both, so that the gradient of `W1` is assembled once
    (one launch, hd_edge_prep; round 5 - three strided copies before) instead of through autograd's slice / cat / transpose nodes (a zero-filled [H, 2H + 2] tensor, a copy
    and an accumulation per slice: ~10 small launches per edge layer)."""

    @staticmethod
    def forward(ctx, dyn, topo, tr, coord, h, W1, b1, zero_h, x4, x04, W2, b2, wa, ba):
        lib = _lib.load()
        H = tr.H
        W1d = W1.detach().contiguous()
        hd = _rows(h.detach())
        # [W1a ; W1b] [2H, H], [b1 | 0] [2H] and the two distance columns [2, H] in one launch (hd_edge_prep)
        Wst = torch.empty((2 * H, H), device=W1d.device, dtype=torch.float32)
        bst = torch.empty((2 * H,), device=W1d.device, dtype=torch.float32)
        wrd = torch.empty((2, H), device=W1d.device, dtype=torch.float32)
        _lib.check(lib.hd_edge_prep(_dev_index(W1d.device), H, 0, W1d.data_ptr(), b1.detach().contiguous().data_ptr(), Wst.data_ptr(),
                                    bst.data_ptr(), wrd.data_ptr(), _stream(W1d.device)), "hd_edge_prep")
        AB = _linear_fwd(hd, Wst, bst)
        x4, x04, W2, b2, wa = (v.detach().contiguous() for v in (x4, x04, W2, b2, wa))
        ba_c = None if ba is None else ba.detach().contiguous()
        out = torch.empty((max(1, tr.M), 4 if coord else tr.H), device=AB.device, dtype=torch.float32)
        prec = _PREC_CODE[getattr(dyn, "training_precision", "fp32")]
        if tr.H < 128:
            prec = 0        # the split arithmetics exist from width 128 up: narrower layers run the exact-fp32 kernels, as in sampling
        # keep W2 P + b2 of every edge row for the backward pass where the library says it pays (large batches) and a gradient
        # will be asked for: [rows, H] fp32 per edge layer, 4.1 GB for the 18 layers of the headline shape at B = 256
        pre2 = None
        if getattr(dyn, "keep_edge_activations", True) and any(ctx.needs_input_grad):
            rows = int(lib.hd_edge_layer_save_rows(dyn._handle(), topo.ptr, prec))
            if rows > 0:
                # [edge rows, H] fp32 per edge layer, alive until the backward pass: B N^2 H layers bytes x 4 in total (4.1 GB at
                # the headline shape; a pocket model with a few hundred nodes reaches tens of GB).  A layer whose buffer does not
                # fit is not an error: it recomputes in its backward pass, like every layer did before round 5
                try:
                    pre2 = torch.empty((rows, tr.H), device=AB.device, dtype=torch.float32)
                except torch.cuda.OutOfMemoryError:
                    pre2 = None
                    if "keep-oom" not in _WARNED:
                        _WARNED.add("keep-oom")
                        import warnings
                        warnings.warn(f"keep_edge_activations: {rows * tr.H * 4 / 2**20:.0f} MiB for one edge layer's pre-activations do "
                                      "not fit; this and later such layers recompute them in the backward pass (same gradients)")
        if prec == 3 and pre2 is None and any(ctx.needs_input_grad):
            prec = 0        # the fp16x3 backward exists on top of the kept pre2 only: this layer runs in exact fp32
            if "fp16x3-fallback" not in _WARNED:
                _WARNED.add("fp16x3-fallback")
                import warnings
                warnings.warn('training_precision "fp16x3": an edge layer whose pre-activations are not kept (keep_edge_activations off, '
                              "out of the memory budget, or a batch small enough for the column-split edge kernel) runs in exact fp32")
        _lib.check(lib.hd_edge_layer_forward_s(dyn._handle(), topo.ptr, int(coord), prec, AB.data_ptr(), x4.data_ptr(),
                                               x04.data_ptr(), wrd.data_ptr(), W2.data_ptr(), b2.data_ptr(), wa.data_ptr(),
                                               None if ba_c is None else ba_c.data_ptr(), out.data_ptr(),
                                               None if pre2 is None else pre2.data_ptr(), _stream(AB.device)),
                   "hd_edge_layer_forward_s")
        ctx.save_for_backward(AB, x4, x04, wrd, W2, b2, wa, hd, Wst, *([] if ba_c is None else [ba_c]), *([] if pre2 is None else [pre2]))
        ctx.misc = (dyn, topo, tr, coord, ba_c is not None, pre2 is not None, prec)
        return out[:tr.M]

    @staticmethod
    def backward(ctx, gout):
        dyn, topo, tr, coord, has_ba, has_pre2, prec = ctx.misc
        saved = ctx.saved_tensors
        AB, x4, x04, wrd, W2, b2, wa, hd, Wst = saved[:9]
        ba = saved[9] if has_ba else None
        pre2 = saved[-1] if has_pre2 else None
        lib = _lib.load()
        ws = tr.workspace()
        dev = AB.device
        M = max(1, tr.M)
        if tr.M == M:
            g = gout.contiguous()
        else:
            g = torch.zeros((M, gout.shape[1]), device=dev, dtype=torch.float32)
        dAB = torch.empty((M, 2 * tr.H), device=dev, dtype=torch.float32)
        dx = torch.empty((M, 4), device=dev, dtype=torch.float32)
        dx0 = torch.empty((M, 4), device=dev, dtype=torch.float32)
        f16 = prec == 3 and tr.H in (128, 256)
        f16ws, n_wg = None, 0
        if f16:
            nw = C.c_int(0)
            f16ws = _f16_workspace(dev, int(lib.hd_edge_layer_f16ws_floats(dyn._handle(), topo.ptr, C.byref(nw))))
            n_wg = nw.value
        _lib.check(lib.hd_edge_layer_backward_s(
            dyn._handle(), topo.ptr, int(coord), prec, AB.data_ptr(), x4.data_ptr(), x04.data_ptr(), wrd.data_ptr(), W2.data_ptr(),
            b2.data_ptr(), wa.data_ptr(), None if ba is None else ba.data_ptr(), g.data_ptr(),
            None if pre2 is None else pre2.data_ptr(), None if f16ws is None else f16ws.data_ptr(), ws["G2"].data_ptr(),
            ws["P"].data_ptr(), ws["G1"].data_ptr(), ws["escal"].data_ptr(), ws["colpart"].data_ptr(), ws["bapart"].data_ptr(),
            ws["b2part"].data_ptr(), ws["wrdpart"].data_ptr(), dAB.data_ptr(), dx.data_ptr(), dx0.data_ptr(), _stream(dev)),
            "hd_edge_layer_backward_s")
        # the one dense reduction over all edge rows: dL/dW2[c][k] = sum_e G2[e][c] P[e][k] (K = edge rows, split-K in slab order)
        if f16:
            # fp16x3: two FP16 pieces per operand, each ranged by one power of two from the maxima the backward stages left in f16ws
            dW2 = torch.empty((tr.H, tr.H), device=dev, dtype=torch.float32)
            slabs = max(1, min(256, tr.rows // 128))
            w6 = _splitk_workspace(dev, slabs * tr.H * tr.H)
            _lib.check(lib.hd_dw2_f16(_dev_index(dev), tr.rows, tr.H, ws["G2"].data_ptr(), ws["P"].data_ptr(), f16ws.data_ptr() + 16,
                                      f16ws.data_ptr() + 16 + 4 * n_wg, n_wg, dW2.data_ptr(), tr.H, w6.data_ptr(), slabs * tr.H * tr.H,
                                      _stream(dev)), "hd_dw2_f16")
        else:
            dW2, _ = _linear_dw(ws["G2"], ws["P"], False, rows=tr.rows)                                       # [H, H]
        # everything else left the kernels as per-tile partial sums: one two-launch column sum over the four arrays
        H = tr.H
        red = torch.empty(4 * H + 1, device=dev, dtype=torch.float32)
        db2, dwrd, dwa, dba = red[:H], red[H:3 * H].view(2, H), red[3 * H:4 * H], red[4 * H:]
        srcs = [ws["b2part"], ws["wrdpart"], ws["colpart"]] + ([ws["bapart"]] if has_ba else [])
        dsts = [db2, dwrd, dwa] + ([dba] if has_ba else [])
        widths = [H, 2 * H, H] + ([1] if has_ba else [])
        n = len(srcs)
        csws = torch.empty(32 * sum(widths), device=dev, dtype=torch.float32)
        _lib.check(lib.hd_colsum_f32(_dev_index(dev), tr.tiles, n, (C.c_void_p * n)(*[t.data_ptr() for t in srcs]),
                                     (C.c_int * n)(*widths), (C.c_void_p * n)(*[t.data_ptr() for t in dsts]),
                                     csws.data_ptr(), _stream(dev)), "hd_colsum_f32")
        if not has_ba:
            dba = None
        # through the first edge Linear: dh = dAB [W1a ; W1b], d[W1a ; W1b] = dAB^T h (+ db1), and W1's gradient in its own layout
        gAB = dAB[:tr.M]
        dh = _linear_dx(gAB, Wst)
        dWst, dbst = _linear_dw(gAB, hd, True)
        dW1 = torch.empty((H, 2 * H + 2), device=dev, dtype=torch.float32)
        dwrd_c = dwrd.contiguous()
        _lib.check(lib.hd_edge_prep(_dev_index(dev), H, 1, dW1.data_ptr(), None, dWst.data_ptr(), None, dwrd_c.data_ptr(), _stream(dev)),
                   "hd_edge_prep")
        return (None, None, None, None, dh, dW1, dbst[:H], None, dx[:tr.M], dx0[:tr.M], dW2, db2, dwa, dba)


def dynamics_forward_train(dyn: EGNN_dynamics_QM9, t, xh, node_mask, edge_mask, context, mol_shape=None) -> torch.Tensor:
    """Differentiable `EGNN_dynamics_QM9._forward` (en_dynamics.py:49-122): same value as the inference path
    (hd_egnn_forward, fp32 mode) up to fp32 round-off of the node-level GEMMs, with gradients to every parameter of
    `dyn.egnn` and to `xh`."""
    if xh.device.type != "cuda":
        raise _lib.HierDiffHipError("training runs only on an MI355X (no CPU fallback)")
    if dyn.precision != "fp32":
        raise _lib.HierDiffHipError('autograd is recording and the dynamics has trainable parameters: the differentiable '
                                    'path uses the exact-fp32 kernels - set dynamics.precision = "fp32", or wrap '
                                    'inference calls in torch.no_grad()')
    dev = xh.device
    B, N, D = xh.shape
    cfg = dyn._cfg
    H, L, S = cfg.hidden_nf, cfg.n_layers, cfg.inv_sublayers
    Fdim = D - 3
    topo = dyn.topology(node_mask, edge_mask, B, N)
    tr = _tables(dyn, topo, dev)
    idx, M = tr.index, tr.M
    egnn = dyn.egnn
    nm = node_mask.reshape(B * N).to(torch.float32)[idx].unsqueeze(1)          # [M, 1]
    xh_c = xh.reshape(B * N, D).to(torch.float32)[idx] * nm
    x_in, hfeat = xh_c[:, :3], xh_c[:, 3:]
    cols = [hfeat]
    if dyn.condition_time:
        tt = t.to(dev, torch.float32).reshape(-1)
        cols.append(tt.reshape(1, 1).expand(M, 1) if tt.numel() == 1 else tt[idx // N].unsqueeze(1))
    if dyn.context_node_nf > 0:
        cols.append(context.to(dev, torch.float32).reshape(B * N, dyn.context_node_nf)[idx])
    h = _Linear.apply(torch.cat(cols, dim=1), egnn.embedding.weight, egnn.embedding.bias)
    x4 = F.pad(x_in, (0, 1))
    x04 = x4
    zero_wa = torch.zeros(H, device=dev)
    nm1 = nm.reshape(-1).contiguous()

    def edge_layer(coord, lin0, lin2, wa, ba, xb):
        return _EdgeLayer.apply(dyn, topo, tr, coord, h, lin0.weight, lin0.bias, zero_wa, xb, x04, lin2.weight, lin2.bias, wa, ba)

    for i in range(L):
        blk = getattr(egnn, f"e_block_{i}")
        xb = x4
        for j in range(S):
            g = getattr(blk, f"gcl_{j}")
            if cfg.attention:
                wa, ba = g.att_mlp[0].weight.reshape(-1), g.att_mlp[0].bias
            else:
                wa, ba = zero_wa, None
            agg = edge_layer(False, g.edge_mlp[0], g.edge_mlp[2], wa, ba, xb)
            h = _NodeMLP.apply(h, agg, g.node_mlp[0].weight, g.node_mlp[0].bias, g.node_mlp[2].weight, g.node_mlp[2].bias, nm1)
        e = blk.gcl_equiv
        xagg = edge_layer(True, e.coord_mlp[0], e.coord_mlp[2], e.coord_mlp[4].weight.reshape(-1), None, xb)
        x4 = (xb + xagg) * nm
        # (the reference's `h = h * node_mask` at the end of a block, egnn_new.py:143: h leaves _NodeMLP already multiplied by the
        # 0 / 1 mask in the GEMM epilogue, so the product is the identity bit for bit - skipped with its backward, 12 launches a step)
        if S == 0:
            h = h * nm
    hout = _Linear.apply(h, egnn.embedding_out.weight, egnn.embedding_out.bias) * nm
    x_final = x4[:, :3]
    if mol_shape is not None:
        fixed = ((idx % N) >= int(mol_shape)).unsqueeze(1)
        x_final = torch.where(fixed, x_in, x_final)
    vel = (x_final - x_in) * nm
    vals = torch.cat([vel, hout[:, :Fdim]], dim=1)             # context, then time columns dropped (en_dynamics.py:99-105)
    out = torch.zeros((B * N, D), device=dev, dtype=torch.float32).index_copy(0, idx, vals).view(B, N, D)
    vel = out[..., :3]
    vel = torch.where(torch.isnan(vel).any(), torch.zeros_like(vel), vel)     # NaN guard without a host sync (:109-111)
    nmf = node_mask.reshape(B, N, 1).to(torch.float32)
    vel = vel - (vel.sum(1, keepdim=True) / nmf.sum(1, keepdim=True)) * nmf
    return torch.cat([vel, out[..., 3:]], dim=2)


# ----------------------------------------------------------------------------- the variational loss around the network call, fused
class _VlbZt(torch.autograd.Function):
    """z_t = alpha(g_t) xh + sigma(g_t) eps (diffusion_qm9.py:566-571) as one launch per direction (`hd_vlb_zt`); differentiable
    with respect to g_t [B] - the learned schedule is part of the graph - not to the data or the noise."""

    @staticmethod
    def forward(ctx, xh, eps, gt):
        lib = _lib.load()
        B = xh.shape[0]
        zt = torch.empty_like(xh)
        _lib.check(lib.hd_vlb_zt(_dev_index(xh.device), B, xh.numel() // B, xh.data_ptr(), eps.data_ptr(), gt.data_ptr(), zt.data_ptr(),
                                 None, None, _stream(xh.device)), "hd_vlb_zt")
        ctx.save_for_backward(xh, eps, gt)
        return zt

    @staticmethod
    def backward(ctx, dzt):
        xh, eps, gt = ctx.saved_tensors
        lib = _lib.load()
        B = xh.shape[0]
        dgt = torch.empty_like(gt)
        dz = dzt.contiguous()
        _lib.check(lib.hd_vlb_zt(_dev_index(xh.device), B, xh.numel() // B, xh.data_ptr(), eps.data_ptr(), gt.data_ptr(), None,
                                 dz.data_ptr(), dgt.data_ptr(), _stream(xh.device)), "hd_vlb_zt")
        return None, None, dgt


class _VlbLoss(torch.autograd.Function):
    """compute_loss of the reference in training mode (t0_always = False) from the network output on: error, SNR weight, KL to the
    prior, log constants, the t = 0 likelihood, the estimator - `hd_vlb_loss_forward` / `_backward`, one launch each (the torch ops
    they replace: ~350 launches of [B, N, 11] tensors per step).  Differentiable with respect to the network output, z_t and the four
    schedule values."""

    @staticmethod
    def forward(ctx, net, zt, gam, xh, eps, nm, t_int, consts):
        lib = _lib.load()
        B, N, D = net.shape
        net, zt, gam = net.contiguous(), zt.contiguous(), gam.contiguous()
        loss = torch.empty(B, device=net.device, dtype=torch.float32)
        err = torch.empty(B, device=net.device, dtype=torch.float32)
        int_nf, cont_nf, l2, T, nv2, nb2, log_nv0 = consts
        _lib.check(lib.hd_vlb_loss_forward(_dev_index(net.device), B, N, D, int_nf, cont_nf, int(l2), T, nv2, nb2, log_nv0, net.data_ptr(),
                                           zt.data_ptr(), xh.data_ptr(), eps.data_ptr(), nm.data_ptr(), gam.data_ptr(), t_int.data_ptr(),
                                           loss.data_ptr(), err.data_ptr(), _stream(net.device)), "hd_vlb_loss_forward")
        ctx.save_for_backward(net, zt, gam, xh, eps, nm, t_int)
        ctx.consts = consts
        ctx.mark_non_differentiable(err)
        return loss, err

    @staticmethod
    def backward(ctx, gout, _gerr):
        net, zt, gam, xh, eps, nm, t_int = ctx.saved_tensors
        lib = _lib.load()
        B, N, D = net.shape
        int_nf, cont_nf, l2, T, nv2, nb2, log_nv0 = ctx.consts
        dnet, dzt, dgam = torch.empty_like(net), torch.empty_like(net), torch.empty_like(gam)
        go = gout.contiguous()
        _lib.check(lib.hd_vlb_loss_backward(_dev_index(net.device), B, N, D, int_nf, cont_nf, int(l2), T, nv2, nb2, log_nv0, net.data_ptr(),
                                            zt.data_ptr(), xh.data_ptr(), eps.data_ptr(), nm.data_ptr(), gam.data_ptr(), t_int.data_ptr(),
                                            go.data_ptr(), dnet.data_ptr(), dzt.data_ptr(), dgam.data_ptr(), _stream(net.device)),
                   "hd_vlb_loss_backward")
        return dnet, dzt, dgam, None, None, None, None, None


def vlb_zt(xh, eps, gt):
    return _VlbZt.apply(xh, eps, gt)


def vlb_loss(net, zt, gam, xh, eps, nm, t_int, consts):
    return _VlbLoss.apply(net, zt, gam, xh, eps, nm, t_int, consts)
