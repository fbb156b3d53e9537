"""TEST INFRASTRUCTURE - synthetic inputs for the stage-2 model `Edge_denoise`: random fragment trees in the two batch
formats the reference feeds it (no RDKit, no dataset files):

    ar_batch      what generation/ar_sampling_nosize.py:72-89 `pad_data` hands to `sample_AR` for a beam of partial trees
    train_batch   what data_utils/dataset_denoise.py:133-311 `PadCollate_onehot` hands to `forward` (one growth step per
                  sample: the tree discovered so far, the node to add, the node it attaches to)

Deterministic in `seed` (numpy PCG64), so fixtures store the seed and the expected outputs only.  Uses the oracle's
restatement of the reference's breadth-first edge layering; the fixture generator checks the oracle against the imported
reference on exactly these batches."""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from oracle.edge_denoise_oracle import adj_to_bfs, concat_layers


def random_tree(rng, n: int):
    """Growth order pi (pi[0] = 0, the root) and parent map of a random tree on n nodes; symmetric adjacency."""
    order = [0] + list(rng.permutation(np.arange(1, n)))
    parent = {}
    adj = np.zeros((n, n), np.float32)
    for k in range(1, n):
        p = order[int(rng.integers(0, k))]
        parent[order[k]] = p
        adj[p, order[k]] = adj[order[k], p] = 1
    return [int(v) for v in order], parent, adj


def synthetic_array_dict(seed: int, out_node_nf: int, n_arrays: int = 6, in_node_nf: int = 8):
    """A stand-in for the reference's `array_dict` pickle (models/edge_denoise.py:19-20; conf/model/edge_denoise.yaml points it at
    a dataset artefact): [list of `n_arrays` fragment property signatures - the first in_node_nf - 1 feature columns a node is matched
    against, :255-256 -, list of the vocabulary slice ("softmax space", sorted ids < out_node_nf) each signature normalises over]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sigs = [rng.integers(0, 4, size=in_node_nf - 1).astype(np.float32) for _ in range(n_arrays)]
    spaces = [sorted(int(v) for v in rng.choice(out_node_nf, size=int(rng.integers(4, max(5, out_node_nf // 2))), replace=False))
              for _ in range(n_arrays)]
    return [sigs, spaces]


def _features(rng, n_list, n_pad, stage_list, orders, in_node_nf, context_nf, vocab_size):
    bs = len(n_list)
    width = in_node_nf + context_nf + 1
    feat = torch.zeros(bs, n_pad, width)
    mask = torch.zeros(bs, n_pad, width)
    pos = torch.zeros(bs, n_pad, 3)
    for i, n in enumerate(n_list):
        f = rng.standard_normal((n, width)).astype(np.float32)
        known = set(orders[i][:stage_list[i]])
        f[:, in_node_nf - 1] = [1.0 if v in known else 0.0 for v in range(n)]          # "discovered" flag column
        f[:, -1] = [float(rng.integers(0, vocab_size - 1)) if v in known else float(vocab_size - 1) for v in range(n)]
        feat[i, :n] = torch.from_numpy(f)
        mask[i, :n] = 1
        pos[i, :n] = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32) * 1.5)
    return feat, mask, pos


def ar_batch(seed: int, n_list: List[int], stage_list: List[int], in_node_nf=8, context_nf=0, vocab_size=50, array_dict=None):
    """stage s of a sample = number of nodes already placed: 0 -> nothing discovered (all-zero adjacency), 1 -> the root is
    marked by a self loop (ar_sampling_nosize.py:202), >= 2 -> the tree edges among the first s nodes of the growth order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bs, n_pad = len(n_list), max(n_list)
    trees = [random_tree(rng, n) for n in n_list]
    adj = torch.zeros(bs, n_pad, n_pad)
    emask = torch.zeros(bs, n_pad, n_pad)
    for i, (n, s) in enumerate(zip(n_list, stage_list)):
        order, parent, full = trees[i]
        if s == 1:
            adj[i, 0, 0] = 1
        for v in order[1:s]:
            adj[i, v, parent[v]] = adj[i, parent[v], v] = 1
        emask[i, :n, :n] = 1 - torch.eye(n)
    feat, mask, pos = _features(rng, n_list, n_pad, stage_list, [t[0] for t in trees], in_node_nf, context_nf, vocab_size)
    if array_dict is not None:
        # sample_AR matches a node's first in_node_nf - 1 feature columns against the signatures (:255-256, check_array_in_list):
        # two nodes in three carry an exact signature (the `diff == 0` return), the others stay random (nearest signature)
        sigs = array_dict[0]
        for i, n in enumerate(n_list):
            for v in range(n):
                if rng.random() < 0.67:
                    feat[i, v, :in_node_nf - 1] = torch.from_numpy(sigs[int(rng.integers(0, len(sigs)))])
    return {'node_feat': [feat, mask], 'node_pos': pos, 'search_adj_matrix': adj, 'edge_mask': emask}


def train_batch(seed: int, n_list: List[int], stage_list: List[int], in_node_nf=8, context_nf=0, vocab_size=50, array_dict=None):
    """One growth step per sample: the first s >= 1 nodes of the growth order are placed, node order[s] is added next."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bs, n_pad = len(n_list), max(n_list)
    trees = [random_tree(rng, n) for n in n_list]
    org = torch.zeros(bs, n_pad, n_pad, dtype=torch.bool)
    search = torch.zeros(bs, n_pad, n_pad, dtype=torch.bool)
    emask = torch.zeros(bs, n_pad, n_pad, dtype=torch.bool)
    predict_idx, last_ind, focal, focal_cand, undiscovered = [], [], [], [], []
    label = torch.zeros(bs, dtype=torch.long)
    for i, (n, s) in enumerate(zip(n_list, stage_list)):
        order, parent, full = trees[i]
        assert 1 <= s < n
        for v in order[1:s]:
            org[i, v, parent[v]] = org[i, parent[v], v] = True
        new = order[s]
        search[i] = org[i]
        search[i, new, parent[new]] = search[i, parent[new], new] = True
        emask[i, :n, :n] = ~torch.eye(n, dtype=torch.bool)
        predict_idx.append(new)
        last_ind.append(parent[new])
        discover = sorted(int(v) for v in org[i].sum(1).nonzero().reshape(-1))
        missing = set(int(v) for v in (torch.from_numpy(full) - org[i, :n, :n].float()).sum(1).nonzero().reshape(-1))
        focal_cand.extend(v + i * n_pad for v in discover)
        focal.extend(v + i * n_pad for v in discover if v in missing)
        undiscovered.append(order[s:])
        label[i] = int(rng.integers(0, vocab_size - 1))
    nums = [int(n) for n in n_list]
    if org.sum() > 0:
        flat = concat_layers([org[i, :nums[i], :nums[i]].nonzero().T.tolist() for i in range(bs)], n_pad)
        flat = [torch.tensor([v for sub in flat[0] for v in sub]), torch.tensor([v for sub in flat[1] for v in sub])]
        orig = concat_layers([adj_to_bfs(org[i, :nums[i], :nums[i]], last_ind[i]) for i in range(bs)], n_pad)
    else:
        flat, orig = [torch.tensor([]), torch.tensor([])], []
    pad = concat_layers([adj_to_bfs(search[i, :nums[i], :nums[i]], predict_idx[i]) for i in range(bs)], n_pad)
    feat, mask, pos = _features(rng, n_list, n_pad, stage_list, [t[0] for t in trees], in_node_nf, context_nf, vocab_size)
    node_array = torch.zeros(bs, n_pad, dtype=torch.long)
    if array_dict is not None:
        # the dataset's per-node signature index (dataset_denoise.py `node_array`); the label of the node to add lies inside the
        # vocabulary slice of its signature (the reference does `softmax_space.index(label)`, :220)
        node_array = torch.from_numpy(rng.integers(0, len(array_dict[0]), size=(bs, n_pad))).long()
        for i in range(bs):
            space = array_dict[1][int(node_array[i, predict_idx[i]])]
            label[i] = int(space[int(rng.integers(0, len(space)))])
    return {'node_feat': [feat, mask.bool()], 'node_array': node_array, 'node_pos': pos,
            'focal': torch.tensor([1 if f in set(focal) else 0 for f in focal_cand]), 'focal_cand': focal_cand,
            'real_focal': [l + i * n_pad for i, l in enumerate(last_ind) if l >= 0],
            'edge_search_pad': pad, 'edge_search_pad_orig': orig, 'edge_search_flat': flat,
            'search_adj_matrix': org, 'edge_mask': emask, 'predict_idx': predict_idx, 'label': label,
            'undiscovered': undiscovered}
