"""Stage-2 model Edge_denoise (/root/reference/models/edge_denoise.py): the oracle against golden vectors generated from the
imported reference module (CPU tier) and the HIP path - hierdiff_amd.edge_denoise.Edge_denoise over hd_egcl_forward / hd_linear -
against the same vectors (GPU tier).

Where the fixtures come from (oracle/make_golden_stage2.py, build container only).  OUTPUTS: the reference class
`models.edge_denoise.Edge_denoise` itself, imported from /root/reference; its module imports three pure-Python graph helpers
(`bfs_node`, `get_bfs_order*`, `get_dfs_order`) from data_utils/data_diffuse.py, a file whose other imports (RDKit, biopandas)
do not exist in this image, so exactly those definitions are executed from the reference's own source text (AST extraction) -
the reference's code, not a restatement.  INPUTS: synthetic, regenerated here from the fixture's seeds by
oracle/edge_denoise_batches.py (random fragment trees in the two batch formats the reference feeds the model) - the reference's
own batches come from data_utils/dataset_denoise.py, which needs RDKit and the dataset files.  Fixtures f17 / f18 `*_ctx_*` carry
context columns in the node features (context_nf = 1), `*_array_*` the vocabulary-slicing `array_dict` (full_softmax off,
edge_denoise.py:214-222, 255-256, 408-417) as a synthetic pickle (edge_denoise_batches.synthetic_array_dict)."""
import copy

import numpy as np
import pytest
import torch

from oracle import edge_denoise_oracle as edo
from oracle import egnn_oracle as orc
from oracle.edge_denoise_batches import ar_batch, train_batch
from tests.helpers import assert_parity, load

AR = ["f18_ar_empty_h64", "f18_ar_roots_h64", "f18_ar_mixed_h64", "f18_ar_mixed_h256", "f18_ar_ctx_h64", "f18_ar_array_h64",
      "f18_ar_array_ctx_h64"]
FWD = ["f17_fwd_h64", "f17_fwd_first_edges_h64", "f17_fwd_h256", "f17_fwd_ctx_h64", "f17_fwd_array_h64", "f17_fwd_array_ctx_h64"]
KEYS = ['focal_loss', 'focal_accuracy', 'edge_loss', 'edge_accuracy', 'node_loss', 'node_accuracy', 'total_loss']


def _ctx(fx):
    return int(fx["context_nf"]) if "context_nf" in fx else 0


def _array_dict(fx):
    """The synthetic [signatures, vocabulary slices] pair of an `*_array_*` fixture, None otherwise."""
    from oracle.edge_denoise_batches import synthetic_array_dict
    seed = int(fx["array_seed"]) if "array_seed" in fx else -1
    return None if seed < 0 else synthetic_array_dict(seed, int(fx["out_node_nf"]))


def _kw(fx):
    return dict(vocab_size=int(fx["vocab_size"]), in_node_nf=8, hidden_nf=int(fx["hidden_nf"]), out_node_nf=int(fx["out_node_nf"]),
                context_nf=_ctx(fx))


def _ar_batch(fx):
    return ar_batch(int(fx["batch_seed"]), [int(v) for v in fx["n_list"]], [int(v) for v in fx["stage_list"]], context_nf=_ctx(fx),
                    vocab_size=int(fx["vocab_size"]), array_dict=_array_dict(fx))


def _train_batch(fx):
    return train_batch(int(fx["batch_seed"]), [int(v) for v in fx["n_list"]], [int(v) for v in fx["stage_list"]], context_nf=_ctx(fx),
                       vocab_size=int(fx["vocab_size"]), array_dict=_array_dict(fx))


def _picked(fx):
    return [[int(v) for v in row if v >= 0] for row in fx["picked"]]


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
    ad = _array_dict(fx)
    res = edo.sample_ar(orc.as_torch_sd(_weights(fx)), _cfg(fx), _ar_batch(fx), array_dict=ad)
    if ad is not None:
        er, npred, picked, adj = res
        assert picked == _picked(fx)
    else:
        er, npred, adj = res
    assert er == _edges(fx)
    assert np.array_equal(adj.numpy(), fx["adj_matrix"])
    assert_parity(npred.numpy(), fx["node_predict"], name, 2e-6, 2e-5)


@pytest.mark.parametrize("name", FWD)
def test_oracle_forward_matches_reference(name):
    fx = load(name)
    out = edo.forward(orc.as_torch_sd(_weights(fx)), _cfg(fx), _train_batch(fx), array_dict=_array_dict(fx))
    for k in KEYS:
        assert abs(float(out[k]) - float(fx[k])) <= 2e-5 * max(1.0, abs(float(fx[k]))), k


def _module(fx, tmp_path=None):
    """The drop-in module built like the reference builds its own: `array_dict` is the PATH of a pickle (edge_denoise.py:19-20)."""
    import pickle
    from hierdiff_amd.edge_denoise import Edge_denoise
    ad, path = _array_dict(fx), None
    if ad is not None:
        path = str(tmp_path / "array_dict.pkl")
        with open(path, "wb") as fh:
            pickle.dump(ad, fh)
    m = Edge_denoise(array_dict=path, full_softmax=ad is None, focal_loss=5, edge_loss=1, node_loss=2, **_kw(fx))
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
    with torch.enable_grad(), pytest.raises(RuntimeError, match="VALUES only"):     # training mode + autograd recording: there is
        m.train()(train_batch(2, [4, 5], [2, 3], vocab_size=int(fx["vocab_size"])))    # no differentiable loss to hand back
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
def test_hip_sample_ar_golden(name, tmp_path):
    """hierdiff_amd.Edge_denoise.sample_AR on the GPU: the same discrete decisions (focal node, attachment, updated adjacency,
    vocabulary slice of the chosen node when an array_dict is loaded) as the reference and its type logits within the
    per-forward bar (rel-L2 1e-4): 8 - 10 E_GCL layers deep."""
    fx = load(name)
    m = _module(fx, tmp_path).to("cuda:0")
    batch = _ar_batch(fx)
    batch = {k: ([t.to("cuda:0") for t in v] if isinstance(v, list) else v.to("cuda:0")) for k, v in batch.items()}
    for rep in range(2):                       # the second call runs on cached graphs
        res = m.sample_AR(copy.deepcopy(batch))
        if "picked" in fx:
            er, npred, picked, adj = res
            assert [list(map(int, p)) for p in picked] == _picked(fx)
        else:
            er, npred, adj = res
        assert er == _edges(fx)
        assert np.array_equal(adj.cpu().numpy(), fx["adj_matrix"])
        assert_parity(npred.cpu().numpy(), fx["node_predict"], name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FWD)
def test_hip_forward_golden(name, tmp_path):
    fx = load(name)
    m = _module(fx, tmp_path).to("cuda:0")
    batch = _train_batch(fx)
    with torch.enable_grad(), pytest.raises(RuntimeError, match="VALUES only"):   # training mode + autograd: no loss without a
        m(batch)                                                                   # grad_fn
    out = m(batch)                             # the tier's default: torch.no_grad(), like the reference's validation loop
    m.eval()
    with torch.enable_grad():                  # ... and an eval-mode call needs no no_grad()
        out_eval = m(batch)
    assert all(float(out[k]) == float(out_eval[k]) for k in KEYS)
    for k in KEYS:
        ref = float(fx[k])
        assert abs(float(out[k]) - ref) <= 1e-4 * max(1.0, abs(ref)), (k, float(out[k]), ref)
