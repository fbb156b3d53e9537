"""Stage-2 model `Edge_denoise` on MI355X - drop-in for /root/reference/models/edge_denoise.py:14-535 (inference values).

The autoregressive decoder that turns a sampled coarse point set into a fragment tree: embeddings, three chains of E_GCL
layers over a dense graph (gcl_full), over the edges discovered so far (gcl_focal) and along breadth-first layers towards the
focal / the new node (gcl_edge, gcl_denoise), and three small prediction heads (focal node, attachment node, fragment type).
Same constructor, same `state_dict` keys, same `forward(batch)` (loss / accuracy values of a training batch) and
`sample_AR(batch)` (one growth step for a batch of partial trees: what generation/ar_sampling_nosize.py:147 calls) as the
reference.  The arithmetic runs in libhierdiff_hip.so: every E_GCL layer through `hd_egcl_forward` (hierdiff_amd.stage2), every
dense layer (embeddings, heads) through `hd_linear`; torch moves memory only (gathers, concatenations, the vocabulary lookup)
and reduces the handful of scalars of the loss.  The tree bookkeeping between the chains (breadth-first edge layers, argmax
over candidates, adjacency updates) is host Python in the reference and here.  Value only (no autograd); no CPU fallback.
"""
from __future__ import annotations

import hashlib
import math
import pickle
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import HierDiffHipError
from .stage2 import E_GCL, egcl_param_shapes


def edge_denoise_param_shapes(vocab_size: int, in_node_nf: int, hidden_nf: int, out_node_nf: int, context_nf: int = 0,
                              in_edge_nf: int = 1, n_layers_full: int = 3, n_layers_focal: int = 3) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameters in the reference's registration order (edge_denoise.py:29-57)."""
    H = hidden_nf
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["feature_embedding.weight"] = (H, in_node_nf); s["feature_embedding.bias"] = (H,)
    s["vocab_embedding.weight"] = (vocab_size, H)
    s["edge_embedding.weight"] = (H, in_edge_nf + 1); s["edge_embedding.bias"] = (H,)
    s["node_embedding.weight"] = (H, 2 * H); s["node_embedding.bias"] = (H,)
    for i in range(n_layers_full):
        for k, v in egcl_param_shapes(H, H, context_nf, True, True).items():
            s[f"gcl_full_{i}.{k}"] = v
    for i in range(n_layers_focal):
        for k, v in egcl_param_shapes(H, H, context_nf, False, True).items():
            s[f"gcl_focal_{i}.{k}"] = v
    for name in ("gcl_edge", "gcl_denoise"):
        for k, v in egcl_param_shapes(H, 1, context_nf, False, False).items():
            s[f"{name}.{k}"] = v
    s["focal_predict.0.weight"] = (H, H + context_nf + 1); s["focal_predict.0.bias"] = (H,)
    s["focal_predict.2.weight"] = (1, H); s["focal_predict.2.bias"] = (1,)
    s["edge_predict.0.weight"] = (H, 3 * H + 1 + 2 * context_nf); s["edge_predict.0.bias"] = (H,)
    s["edge_predict.2.weight"] = (1, H); s["edge_predict.2.bias"] = (1,)
    s["node_predict.0.weight"] = (H, H + context_nf); s["node_predict.0.bias"] = (H,)
    s["node_predict.2.weight"] = (out_node_nf, H); s["node_predict.2.bias"] = (out_node_nf,)
    return s


def synthetic_edge_denoise_state_dict(seed: int = 0, coord_gain: float = 0.3, **shape_kw) -> "OrderedDict[str, np.ndarray]":
    """Deterministic nn.Linear / nn.Embedding-style weights keyed by tensor name (the reference ships no stage-2 checkpoint)."""
    shapes = edge_denoise_param_shapes(**shape_kw)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        digest = hashlib.sha256(f"edge_denoise:{seed}:{name}".encode()).digest()
        rng = np.random.Generator(np.random.PCG64(int.from_bytes(digest[:8], "little")))
        if name == "vocab_embedding.weight":
            out[name] = rng.standard_normal(shape).astype(np.float32)
            continue
        wshape = shapes[name[:-4] + "weight"] if name.endswith("bias") else shape
        bound = 1.0 / math.sqrt(wshape[1])
        if name.endswith("coord_mlp.2.weight"):
            bound = coord_gain * math.sqrt(6.0 / (shape[0] + shape[1]))
        out[name] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def bfs_layers(edges, n_nodes: int, start: int) -> List[List[List[int]]]:
    """Layers of [child, parent] pairs reached breadth-first from `start` over the directed pairs `edges`, farthest layer
    first (data_utils/data_diffuse.py:60-79 `get_bfs_order_new`).  Like the reference it does not terminate on a graph whose
    `n_nodes` nodes are not all reachable from `start`; callers pass connected trees.  Per layer the reference scans ALL pairs
    and files [b, a] for every pair whose source is already seen and whose target is not (a target reached from two sources of
    the same layer is listed twice); the same pairs in the same edge-list order come out here."""
    pairs = [(int(a), int(b)) for a, b in edges]
    seen = {int(start)}
    layers: List[List[List[int]]] = []
    while len(seen) < n_nodes:
        layer = [[b, a] for a, b in pairs if a in seen and b not in seen]
        if not layer:
            raise ValueError("bfs_layers: the edge set is not connected to `start` (the reference loops forever here)")
        seen.update(b for b, _ in layer)
        layers.append(layer)
    layers.reverse()
    return layers


class Edge_denoise(nn.Module):
    """HIP implementation of models/edge_denoise.py:14-535."""

    def __init__(self, vocab_size, in_node_nf, hidden_nf, out_node_nf, array_dict, context_nf=0, in_edge_nf=1,
                 n_layers_full=3, n_layers_focal=3, focal_loss=1, edge_loss=1, node_loss=1, perturb_loss=1, full_softmax=False):
        super().__init__()
        if not full_softmax:
            with open(array_dict, 'rb') as fh:
                self.array_dict = pickle.load(fh)
        else:
            self.array_dict = None
        H = hidden_nf
        self.in_node_nf, self.hidden_nf, self.context_nf = in_node_nf, H, context_nf
        self.n_layers_full, self.n_layers_focal = n_layers_full, n_layers_focal
        self.feature_embedding = nn.Linear(in_node_nf, H)
        self.vocab_embedding = nn.Embedding(vocab_size, H)
        self.edge_embedding = nn.Linear(in_edge_nf + 1, H)
        self.node_embedding = nn.Linear(2 * H, H)
        kw = dict(context_nf=context_nf, act_fn=nn.SiLU(), recurrent=True, tanh=True, coords_range=30, agg='sum', coord_update=True)
        for i in range(n_layers_full):
            self.add_module("gcl_full_%d" % i, E_GCL(H, H, H, edges_in_d=H, attention=True, edge_update=True, **kw))
        for i in range(n_layers_focal):
            self.add_module("gcl_focal_%d" % i, E_GCL(H, H, H, edges_in_d=H, attention=False, edge_update=True, **kw))
        self.add_module("gcl_edge", E_GCL(H, H, H, edges_in_d=1, attention=False, edge_update=False, **kw))
        self.add_module("gcl_denoise", E_GCL(H, H, H, edges_in_d=1, attention=False, edge_update=False, **kw))
        self.focal_predict = nn.Sequential(nn.Linear(H + context_nf + 1, H), nn.SiLU(), nn.Linear(H, 1), nn.Sigmoid())
        self.edge_predict = nn.Sequential(nn.Linear(3 * H + 1 + 2 * context_nf, H), nn.SiLU(), nn.Linear(H, 1))
        self.node_predict = nn.Sequential(nn.Linear(H + context_nf, H), nn.SiLU(), nn.Linear(H, out_node_nf))
        self.loss_lambda = {'focal_loss': focal_loss, 'edge_loss': edge_loss, 'node_loss': node_loss}
        self._edges_dict: Dict[Tuple[int, int, str], List[torch.Tensor]] = {}

    # ------------------------------------------------------------------ HIP dense layers
    def _device(self) -> torch.device:
        dev = self.feature_embedding.weight.device
        if dev.type != "cuda":
            raise HierDiffHipError("Edge_denoise runs only on an MI355X: move the module to a cuda device (there is no CPU fallback)")
        _lib.require_gpu()
        return dev

    def _linear(self, x: torch.Tensor, layer: nn.Linear, act: int = 0) -> torch.Tensor:
        """act(layer(x)) through hd_linear (act: 0 none, 1 SiLU, 2 sigmoid); x [..., K] -> [..., N]."""
        dev = self._device()
        lead = x.shape[:-1]
        x2 = x.detach().to(dev, torch.float32).reshape(-1, x.shape[-1]).contiguous()
        W = layer.weight.detach().to(torch.float32).contiguous()
        b = None if layer.bias is None else layer.bias.detach().to(torch.float32).contiguous()
        M, K, N = x2.shape[0], x2.shape[1], W.shape[0]
        y = torch.empty((M, N), device=dev, dtype=torch.float32)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(_lib.load().hd_linear(idx, x2.data_ptr(), M, K, K, W.data_ptr(), None if b is None else b.data_ptr(), N, act,
                                         y.data_ptr(), N, torch.cuda.current_stream(dev).cuda_stream), "hd_linear")
        return y.reshape(*lead, N)

    def _head(self, seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
        """Linear + SiLU + Linear (+ Sigmoid) heads (:55-57)."""
        return self._linear(self._linear(x, seq[0], 1), seq[2], 2 if len(seq) == 4 else 0)

    def get_adj_matrix(self, n_nodes, batch_size, device):
        """:507-526: every (i, j) pair incl. i == j per graph, cached per (n_nodes, batch_size)."""
        key = (int(n_nodes), int(batch_size), str(device))
        if key not in self._edges_dict:
            ar = torch.arange(n_nodes)
            r = ar.repeat_interleave(n_nodes).repeat(batch_size)
            c = ar.repeat(n_nodes).repeat(batch_size)
            off = (torch.arange(batch_size) * n_nodes).repeat_interleave(n_nodes * n_nodes)
            self._edges_dict[key] = [(r + off).to(device), (c + off).to(device)]
        return self._edges_dict[key]

    # ------------------------------------------------------------------ shared front end (:83-112 == :275-296)
    def _embed_and_full(self, h, x, adj_note, node_mask, edge_mask, bs, n):
        dev = self._device()
        h = h.to(dev, torch.float32)
        h_f = self._linear(h[:, :self.in_node_nf], self.feature_embedding)
        h_v = self.vocab_embedding.weight.detach()[h[:, self.in_node_nf + self.context_nf].long()]
        hh = self._linear(torch.cat([h_f, h_v], dim=1), self.node_embedding)
        if self.context_nf > 0:
            hh = torch.cat([hh, h[:, self.in_node_nf:self.in_node_nf + self.context_nf]], dim=1)
        edges_full = self.get_adj_matrix(n, bs, dev)
        radial = torch.sum((x[edges_full[0]] - x[edges_full[1]]) ** 2, dim=1, keepdim=True)
        eff = self._linear(torch.cat([radial, adj_note.to(dev, torch.float32)], dim=1), self.edge_embedding)
        for i in range(self.n_layers_full):
            hh, x, eff = self._modules["gcl_full_%d" % i](hh, edges_full, x, edge_attr=eff, node_mask=node_mask, edge_mask=edge_mask)
        return hh, x, eff.view(bs, n, n, -1)

    def _walk(self, name, layers, h, x, node_mask):
        """One E_GCL applied along a list of edge layers with edge_attr = squared length (:155-159, :201-205)."""
        dev = h.device
        host = [torch.tensor(layer, dtype=torch.long).reshape(-1, 2).T.contiguous() for layer in layers]
        both = torch.cat(host, dim=1).to(dev)                 # every layer's pairs in ONE host-to-device copy
        lo = 0
        for e in host:
            k = int(e.shape[1])
            ed = both[:, lo:lo + k]
            lo += k
            ea = torch.sum((x[ed[0]] - x[ed[1]]) ** 2, dim=1, keepdim=True)
            h, x = self._modules[name](h, [e[0], e[1]], x, edge_attr=ea, node_mask=node_mask)
        return h, x

    def _edge_head(self, h, x, eff, focals, bs, n):
        """edge_predict over (focal node, every node of its graph) (:162-173 == :345-356)."""
        fidx = torch.tensor(focals, dtype=torch.long, device=h.device)
        hv, xv = h.view(bs, n, -1), x.view(bs, n, -1)
        h_focal = h[fidx].unsqueeze(1).expand(-1, n, -1)
        x_focal = x[fidx].unsqueeze(1)
        edge_focal = eff[fidx // n, fidx % n]
        h_att, x_att = hv[fidx // n], xv[fidx // n]
        dist = torch.sum((x_att - x_focal) ** 2, dim=2, keepdim=True)
        return self._head(self.edge_predict, torch.cat([h_focal, edge_focal, h_att, dist], dim=-1))

    # ------------------------------------------------------------------ reference API
    def forward(self, batch):
        """:61-248: {'focal_loss', 'focal_accuracy', 'edge_loss', 'edge_accuracy', 'node_loss', 'node_accuracy', 'total_loss'}
        of a training batch.  VALUES ONLY (inference / evaluation): the reference's `forward(batch)` is also its stage-2 training
        objective (trainmodule/train_edge_denoise_pl.py backpropagates `total_loss`); the HIP layers under this module have no
        backward, so a call that could be asked for gradients - training mode with autograd recording - raises instead of
        returning a loss without a grad_fn.  Call it under `torch.no_grad()` or after `.eval()`."""
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("hierdiff_amd.Edge_denoise.forward computes loss VALUES only (no backward through the HIP stage-2 "
                               "layers): call it under torch.no_grad() or in eval mode; stage-2 training is out of scope "
                               "(SURVEY.md section 8f row 4)")
        with torch.no_grad():
            return self._forward_values(batch)

    def _forward_values(self, batch):
        dev = self._device()
        h = batch['node_feat'][0]
        bs, n = h.shape[:2]
        array = batch['node_array'] if self.array_dict is not None else None
        x = batch['node_pos'].to(dev, torch.float32).reshape(bs * n, -1)
        predict_idx = [int(p) for p in batch['predict_idx']]
        edge_search = [list(l) for l in batch['edge_search_pad']]
        edge_search_orig = [list(l) for l in batch['edge_search_pad_orig']]
        flat = batch['edge_search_flat']
        node_mask = batch['node_feat'][1][:, :, 0].to(dev, torch.float32).reshape(bs * n, -1)
        edge_mask = batch['edge_mask'].to(dev, torch.float32).reshape(bs * n * n, -1)
        focal = torch.as_tensor(batch['focal']).to(dev)
        focal_cand, real_focal = [int(v) for v in batch['focal_cand']], [int(v) for v in batch['real_focal']]
        undiscovered = [[int(v) for v in u] for u in batch['undiscovered']]
        label = torch.as_tensor(batch['label']).to(dev)
        adj = torch.as_tensor(batch['search_adj_matrix']).to(dev)
        val = torch.sum(adj.reshape(bs * n, n), dim=-1, keepdim=True).to(torch.float32)
        h, x, eff = self._embed_and_full(h.reshape(bs * n, -1), x, adj.reshape(bs * n * n, 1), node_mask, edge_mask, bs, n)
        max_depth = len(edge_search)
        zero = torch.zeros((), device=dev)
        focal_loss, focal_acc = zero, 0.0
        if max_depth > 1:
            e0, e1 = (torch.as_tensor(t).long() for t in flat)
            d0, d1 = e0.to(dev), e1.to(dev)
            ef = eff[d0 // n, d0 % n, d1 % n, :].reshape(e0.shape[0], -1)
            for i in range(self.n_layers_focal):
                h, x, ef = self._modules['gcl_focal_%d' % i](h, [e0, e1], x, edge_attr=ef, node_mask=node_mask)
            cand = torch.tensor(focal_cand, dtype=torch.long, device=dev)
            fp = self._head(self.focal_predict, torch.cat([h[cand], val[cand]], dim=1))
            # split_edges (:502-506) walks `for e in edge_search_flat`, i.e. over the TWO index tensors, and files e[0] of each:
            # only the graphs holding the first source and the first target node (graph 0 in practice) count as having edges,
            # so only their candidates enter the focal loss.  Reproduced as is.
            ew = [0] * bs
            for t in (e0, e1):
                if t.numel() > 0:
                    ew[int(t[0]) // n] += 1
            bins = self.split_nodes(focal_cand, n, bs)
            nw = np.cumsum([0] + [len(b) for b in bins])
            for i in range(bs):
                if ew[i] != 0:
                    focal_loss = focal_loss + nn.functional.binary_cross_entropy(fp[nw[i]:nw[i + 1]].squeeze(-1), focal[nw[i]:nw[i + 1]].float())
            hit, cnt = 0, 0
            fp_host, focal_host = fp.squeeze(-1).cpu(), focal.cpu()
            for i, fk in enumerate(bins):
                if len(fk) > 0:
                    pos = [focal_cand.index(j + i * n) for j in fk]
                    if int(focal_host[pos[int(torch.argmax(fp_host[pos]))]]) == 1:
                        hit += 1
                    cnt += 1
            focal_acc = hit / (cnt + 1e-8)
        circle = [[i * n, i * n] for i in range(bs)]
        h, x = self._walk('gcl_edge', ([circle] + edge_search_orig)[:max_depth], h, x, node_mask)
        edge_loss, edge_acc = zero, 0.0
        if max_depth > 0 and len(real_focal) > 0:
            ep = self._edge_head(h, x, eff, real_focal, bs, n)
            fi, cnt, hit = 0, 0, 0
            for i in range(bs):
                if predict_idx[i] != 0:
                    target = torch.tensor([undiscovered[i].index(predict_idx[i])], device=dev)
                    logits = ep[fi, undiscovered[i], :].squeeze(-1).unsqueeze(0)
                    edge_loss = edge_loss + nn.functional.cross_entropy(logits, target)
                    hit += int(torch.argmax(logits, dim=-1) == target)
                    cnt += 1
                    fi += 1
            edge_acc = hit / (cnt + 1e-8)
        if max_depth > 0:
            h, x = self._walk('gcl_denoise', ([circle] + edge_search)[:max_depth + 1], h, x, node_mask)
        hv = h.view(bs, n, -1)
        h_node = torch.stack([hv[i, predict_idx[i], :] for i in range(bs)])
        node_predict = self._head(self.node_predict, h_node)
        node_loss, hit = zero, 0
        np_host = node_predict.cpu()
        for i in range(bs):
            space = self._softmax_space(array, i, predict_idx[i], bs, n, node_predict.shape[1])
            lab = space.index(int(label[i]))
            node_loss = node_loss + nn.functional.cross_entropy(node_predict[i, space].unsqueeze(0), torch.tensor([lab], device=dev))
            hit += int(int(torch.argmax(np_host[i, space])) == lab)
        total = self.loss_lambda['focal_loss'] * focal_loss + self.loss_lambda['edge_loss'] * edge_loss + \
            self.loss_lambda['node_loss'] * node_loss
        return {'focal_loss': focal_loss, 'focal_accuracy': torch.tensor(focal_acc), 'edge_loss': edge_loss,
                'edge_accuracy': torch.tensor(edge_acc), 'node_loss': node_loss, 'node_accuracy': torch.tensor(hit / bs),
                'total_loss': total}

    def _softmax_space(self, array, i, node, bs, n, width):
        """:214-222: the vocabulary slice a node's type is normalised over (everything when array_dict is None)."""
        if self.array_dict is None:
            return list(range(width))
        return list(self.array_dict[1][int(array.reshape(bs, n)[i, node])])

    def _layers_frozen(self, on: bool):
        """Parameters do not change inside one sample_AR / forward call: the E_GCL layers compare their parameter versions with the
        library's copy once per call instead of once per layer application (~40 per growth step)."""
        for m in self.modules():
            if isinstance(m, E_GCL):
                if on:
                    m._frozen = False
                    m._sync_weights()
                m._frozen = on

    @torch.no_grad()
    def sample_AR(self, batch):
        """:250-420: for every partial tree of the batch choose the focal node, the node to attach to it and the type
        logits of that node.  Returns (edges_result, node_predict[, array], adj_matrix) like the reference."""
        self._device()
        self._layers_frozen(True)
        try:
            return self._sample_ar(batch)
        finally:
            self._layers_frozen(False)

    def _sample_ar(self, batch):
        dev = self._device()
        h = batch['node_feat'][0]
        bs, n = h.shape[:2]
        h = h.reshape(bs * n, -1)
        array = None
        if self.array_dict is not None:
            feats = h.cpu().numpy()
            array = torch.tensor([check_array_in_list(a[:-(2 + self.context_nf)], self.array_dict[0]) for a in feats]).view(bs, n)
        x = batch['node_pos'].to(dev, torch.float32).reshape(bs * n, -1)
        nm_host = batch['node_feat'][1][:, :, 0].detach().cpu().to(torch.float32)
        node_nums = torch.sum(nm_host, dim=1).int().tolist()
        node_mask = nm_host.reshape(bs * n, -1).to(dev)
        edge_mask = batch['edge_mask'].to(dev, torch.float32).reshape(bs * n * n, -1)
        adj = batch['search_adj_matrix'].detach().cpu().to(torch.float32).clone()     # the tree bookkeeping lives on the host
        val = torch.sum(adj.reshape(bs * n, n), dim=-1, keepdim=True).to(dev)
        valid = nm_host.reshape(bs * n).numpy() != 0
        rowsum = adj.reshape(bs * n, n).sum(-1).numpy()
        discovered = [int(i) for i in np.nonzero(valid & (rowsum > 0))[0]]
        undiscovered = [int(i) for i in np.nonzero(valid & (rowsum == 0))[0]]
        adj = adj - torch.diag_embed(torch.diagonal(adj, dim1=1, dim2=2))
        h, x, eff = self._embed_and_full(h, x, adj.reshape(bs * n * n, 1), node_mask, edge_mask, bs, n)
        have_edges = bool(adj.sum() > 0)
        if have_edges:
            per = [adj[i][:node_nums[i], :node_nums[i]].nonzero().T.tolist() for i in range(bs)]
            e0 = torch.tensor([v + i * n for i, p in enumerate(per) for v in p[0]], dtype=torch.long)
            e1 = torch.tensor([v + i * n for i, p in enumerate(per) for v in p[1]], dtype=torch.long)
            d0, d1 = e0.to(dev), e1.to(dev)
            ef = eff[d0 // n, d0 % n, d1 % n, :].reshape(e0.shape[0], -1)
            for i in range(self.n_layers_focal):
                h, x, ef = self._modules['gcl_focal_%d' % i](h, [e0, e1], x, edge_attr=ef, node_mask=node_mask)
            # focal_predict of every discovered node of the beam in ONE head call (the reference calls it per sample, :311-316;
            # a head's rows are independent - one fmaf chain per output in hd_linear - so the scores are the same bits), one
            # device-to-host copy, then the per-sample argmax (first maximum, like torch.argmax) on the host
            bins = self.split_nodes(discovered, n, bs)
            didx = torch.tensor(discovered, dtype=torch.long, device=dev)
            score = self._head(self.focal_predict, torch.cat([h[didx], val[didx]], dim=-1)).reshape(-1).cpu().numpy()
            focal, lo = [], 0
            for i in range(bs):
                k = len(bins[i])
                focal.append(bins[i][int(np.argmax(score[lo:lo + k]))] + i * n if k > 0 else -1)
                lo += k
        elif len(discovered) == 0:
            focal = [-1] * bs
        else:
            focal = [0] * bs
        edges_result = []
        circle = [[i * n, i * n] for i in range(bs)]
        if len(discovered) > 0:
            if have_edges:
                adj_np = adj.numpy()                        # the host tensor's own storage: no per-sample tensor ops below
                per = [self.adj_matrix_to_edges_bfs(adj_np[i, :node_nums[i], :node_nums[i]], None, focal[i] % n) if focal[i] >= 0 else []
                       for i in range(bs)]
                h, x = self._walk('gcl_edge', [circle] + self.concat_edges(per, n), h, x, node_mask)
            fr = [f for f in focal if f >= 0]
            ep = self._edge_head(h, x, eff, fr, bs, n).cpu()
            ubins = self.split_nodes(undiscovered, n, bs)
            fi = 0
            for i in range(bs):
                if 0 not in ubins[i]:
                    end = ubins[i][int(torch.argmax(ep[fi, ubins[i], :]))]
                    a = fr[fi] % n
                    adj[i] = self.attach_to_adj_matrix(adj[i], [[a, end], [end, a]])
                    edges_result.append([a, end])
                    fi += 1
                else:
                    edges_result.append([-1, 0])
        else:
            edges_result = [[-1, 0] for _ in range(bs)]
        adj_np = adj.numpy()
        per = [self.adj_matrix_to_edges_bfs(adj_np[i, :node_nums[i], :node_nums[i]], None, edges_result[i][1]) if focal[i] > 0 else []
               for i in range(bs)]
        h, x = self._walk('gcl_denoise', [circle] + self.concat_edges(per, n), h, x, node_mask)
        hv = h.view(bs, n, -1)
        h_node = torch.stack([hv[i, edges_result[i][1], :] for i in range(bs)])
        node_predict = self._head(self.node_predict, h_node)
        if self.array_dict is not None:
            picked = [self.array_dict[1][int(array[i, edges_result[i][1]])] for i in range(bs)]
        edges_result = [e if e[0] >= 0 else [0] for e in edges_result]
        adj = adj.to(batch['search_adj_matrix'].device)
        if self.array_dict is not None:
            return edges_result, node_predict, picked, adj
        return edges_result, node_predict, adj

    # ------------------------------------------------------------------ host helpers of the reference (:422-505)
    def strip_adj_matrix(self, adj_matrix_pad, n_nodes):
        return adj_matrix_pad[:n_nodes, :n_nodes]

    def adj_matrix_to_edges_flat(self, adj_matrix):
        return adj_matrix.nonzero().T.tolist()

    def adj_matrix_to_edges_bfs(self, adj_matrix, blur_feature, end, priority=False):
        a = adj_matrix.detach().cpu().numpy() if isinstance(adj_matrix, torch.Tensor) else np.asarray(adj_matrix)
        r, c = np.nonzero(a)                                   # row-major: the order of torch.nonzero
        if r.size == 0:
            return [[]]
        r, c = r.tolist(), c.tolist()
        return bfs_layers(zip(r, c), len(set(r) | set(c)), int(end))

    def attach_to_adj_matrix(self, adj_matrix, edges):
        for e in edges:
            adj_matrix[e[0], e[1]] = 1
        return adj_matrix

    def concat_edges(self, edges, n_nodes):
        """Layer l of sample i, node ids offset by i * n_nodes, appended to the batch's layer l (:480-494 + flat_add)."""
        depth = max(len(e) for e in edges)
        out: List[list] = [[] for _ in range(depth)]
        for i, layers in enumerate(edges):
            for li, layer in enumerate(layers):
                if len(layer) > 0 and isinstance(layer[0], (int, np.integer)):
                    out[li].append([int(v) + i * n_nodes for v in layer])
                else:
                    out[li].extend([(int(a) + i * n_nodes, int(b) + i * n_nodes) for a, b in layer])
        return out

    def split_nodes(self, node_idxs, n_nodes, bs):
        bins: List[List[int]] = [[] for _ in range(bs)]
        for i in node_idxs:
            bins[i // n_nodes].append(i % n_nodes)
        return bins

    def split_edges(self, edges, n_nodes, bs):
        bins: List[List[List[int]]] = [[] for _ in range(bs)]
        for e in edges:
            bins[e[0] // n_nodes].append([e[0] % n_nodes, e[1] % n_nodes])
        return bins


def check_array_in_list(array, list_a):
    """:537-546: index of the reference array equal (else nearest in squared distance) to `array`."""
    if isinstance(array, torch.Tensor):
        array = array.cpu().numpy()
    best, best_d = 0, None
    for ind, ref in enumerate(list_a):
        d = float(((array - ref) ** 2).sum())
        if d == 0:
            return ind
        if best_d is None or d < best_d:
            best, best_d = ind, d
    return best
