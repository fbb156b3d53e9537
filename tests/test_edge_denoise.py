"""Stage-2 model Edge_denoise (/root/reference/models/edge_denoise.py): the oracle against golden vectors generated from the
imported reference module (CPU tier) and the HIP path - hierdiff_amd.edge_denoise.Edge_denoise over hd_egcl_forward / hd_linear -
against the same vectors (GPU tier).  Inputs are regenerated from the fixture's seeds (oracle/edge_denoise_batches.py)."""
import copy

import numpy as np
import pytest
import torch

from oracle import edge_denoise_oracle as edo
from oracle import egnn_oracle as orc
from oracle.edge_denoise_batches import ar_batch, train_batch
from tests.helpers import assert_parity, load

AR = ["f18_ar_empty_h64", "f18_ar_roots_h64", "f18_ar_mixed_h64", "f18_ar_mixed_h256"]
FWD = ["f17_fwd_h64", "f17_fwd_first_edges_h64", "f17_fwd_h256"]
KEYS = ['focal_loss', 'focal_accuracy', 'edge_loss', 'edge_accuracy', 'node_loss', 'node_accuracy', 'total_loss']


def _kw(fx):
    return dict(vocab_size=int(fx["vocab_size"]), in_node_nf=8, hidden_nf=int(fx["hidden_nf"]), out_node_nf=int(fx["out_node_nf"]),
                context_nf=0)


def _weights(fx):
    from hierdiff_amd.edge_denoise import synthetic_edge_denoise_state_dict
    return synthetic_edge_denoise_state_dict(int(fx["weight_seed"]), **_kw(fx))


def _cfg(fx):
    return edo.EDCfg(focal_loss=5, edge_loss=1, node_loss=2, **_kw(fx))


def _edges(fx):
    return [[int(v) for v in e if v >= 0] for e in fx["edges_result"]]


@pytest.mark.parametrize("name", AR)
def test_oracle_sample_ar_matches_reference(name):
    fx = load(name)
    batch = ar_batch(int(fx["batch_seed"]), [int(v) for v in fx["n_list"]], [int(v) for v in fx["stage_list"]],
                     vocab_size=int(fx["vocab_size"]))
    er, npred, adj = edo.sample_ar(orc.as_torch_sd(_weights(fx)), _cfg(fx), batch)
    assert er == _edges(fx)
    assert np.array_equal(adj.numpy(), fx["adj_matrix"])
    assert_parity(npred.numpy(), fx["node_predict"], name, 2e-6, 2e-5)


@pytest.mark.parametrize("name", FWD)
def test_oracle_forward_matches_reference(name):
    fx = load(name)
    batch = train_batch(int(fx["batch_seed"]), [int(v) for v in fx["n_list"]], [int(v) for v in fx["stage_list"]],
                        vocab_size=int(fx["vocab_size"]))
    out = edo.forward(orc.as_torch_sd(_weights(fx)), _cfg(fx), batch)
    for k in KEYS:
        assert abs(float(out[k]) - float(fx[k])) <= 2e-5 * max(1.0, abs(float(fx[k]))), k


def _module(fx):
    from hierdiff_amd.edge_denoise import Edge_denoise
    m = Edge_denoise(array_dict=None, full_softmax=True, focal_loss=5, edge_loss=1, node_loss=2, **_kw(fx))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in _weights(fx).items()})
    return m


def test_module_mirrors_reference_layout():
    from hierdiff_amd import _lib
    from hierdiff_amd.edge_denoise import edge_denoise_param_shapes
    fx = load("f18_ar_mixed_h64")
    m = _module(fx)
    shapes = edge_denoise_param_shapes(**_kw(fx))
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert sum(p.numel() for p in m.parameters()) == sum(int(np.prod(s)) for s in shapes.values())
    if _lib.load().hd_device_count() == 0:          # no CPU fallback
        batch = ar_batch(1, [4, 5], [2, 3], vocab_size=int(fx["vocab_size"]))
        with pytest.raises(_lib.HierDiffHipError):
            m.sample_AR(batch)


def test_bfs_layers_restate_the_reference_order():
    """A path 3-1-0-2 plus a branch 1-4, searched from node 4: layers come out farthest first, [child, parent] pairs."""
    from hierdiff_amd.edge_denoise import bfs_layers
    pairs = [(3, 1), (1, 0), (0, 2), (1, 4)]
    edges = np.array([p for a, b in pairs for p in ((a, b), (b, a))])
    edges = edges[np.lexsort((edges[:, 1], edges[:, 0]))]                 # nonzero() order: by row, then column
    assert bfs_layers(edges, 5, 4) == edo.bfs_layers(edges, 5, 4) == [[[2, 0]], [[0, 1], [3, 1]], [[1, 4]]]
    with pytest.raises(ValueError):
        bfs_layers(np.array([(0, 1), (1, 0), (2, 3), (3, 2)]), 4, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", AR)
def test_hip_sample_ar_golden(name):
    """hierdiff_amd.Edge_denoise.sample_AR on the GPU: the same discrete decisions (focal node, attachment, updated adjacency)
    as the reference and its type logits within the per-forward bar (rel-L2 1e-4): 8 - 10 E_GCL layers deep."""
    fx = load(name)
    m = _module(fx).to("cuda:0")
    batch = ar_batch(int(fx["batch_seed"]), [int(v) for v in fx["n_list"]], [int(v) for v in fx["stage_list"]],
                     vocab_size=int(fx["vocab_size"]))
    batch = {k: ([t.to("cuda:0") for t in v] if isinstance(v, list) else v.to("cuda:0")) for k, v in batch.items()}
    for rep in range(2):                       # the second call runs on cached graphs
        er, npred, adj = m.sample_AR(copy.deepcopy(batch))
        assert er == _edges(fx)
        assert np.array_equal(adj.cpu().numpy(), fx["adj_matrix"])
        assert_parity(npred.cpu().numpy(), fx["node_predict"], name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FWD)
def test_hip_forward_golden(name):
    fx = load(name)
    m = _module(fx).to("cuda:0")
    batch = train_batch(int(fx["batch_seed"]), [int(v) for v in fx["n_list"]], [int(v) for v in fx["stage_list"]],
                        vocab_size=int(fx["vocab_size"]))
    out = m(batch)
    for k in KEYS:
        ref = float(fx[k])
        assert abs(float(out[k]) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, float(out[k]), ref)
