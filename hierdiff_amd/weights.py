"""Weight layout of the coarse-grained diffusion model and a deterministic synthetic generator.

The reference ships no checkpoint (endiffusion/conf/sample.yaml:17 points at the author's home
directory), so parity fixtures, tests and the bench all use weights produced by
`synthetic_state_dict`: every tensor is drawn from its own numpy PCG64 stream keyed by the tensor's
state_dict name, so the same weights can be regenerated anywhere without shipping them.

Key layout follows the reference modules verbatim:
  endiffusion/models/layers/egnn_new.py:9-33   GCL            edge_mlp.{0,2}, node_mlp.{0,2}, att_mlp.0
  endiffusion/models/layers/egnn_new.py:74-89  EquivariantUpdate  coord_mlp.{0,2,4}
  endiffusion/models/layers/egnn_new.py:180-190 EGNN          embedding, embedding_out, e_block_{i}
  endiffusion/models/noise_model.py:163-174    GammaNetwork   l1, l2, l3, gamma_0, gamma_1
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np


def dynamics_param_shapes(in_node_nf: int, context_node_nf: int, hidden_nf: int, n_layers: int,
                          inv_sublayers: int, attention: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered (name -> shape) of EGNN_dynamics_QM9's parameters, in the reference's registration order.

    `in_node_nf` already includes the time column when condition_time is set
    (endiffusion/train_module/diffusion_qm9.py:89-90); the EGNN sees in_node_nf + context_node_nf
    input features (endiffusion/models/module/en_dynamics.py:17-18).
    """
    H = hidden_nf
    fin = in_node_nf + context_node_nf
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    shapes["egnn.embedding.weight"] = (H, fin)
    shapes["egnn.embedding.bias"] = (H,)
    shapes["egnn.embedding_out.weight"] = (fin, H)
    shapes["egnn.embedding_out.bias"] = (fin,)
    for i in range(n_layers):
        for j in range(inv_sublayers):
            p = f"egnn.e_block_{i}.gcl_{j}."
            shapes[p + "edge_mlp.0.weight"] = (H, 2 * H + 2)
            shapes[p + "edge_mlp.0.bias"] = (H,)
            shapes[p + "edge_mlp.2.weight"] = (H, H)
            shapes[p + "edge_mlp.2.bias"] = (H,)
            shapes[p + "node_mlp.0.weight"] = (H, 2 * H)
            shapes[p + "node_mlp.0.bias"] = (H,)
            shapes[p + "node_mlp.2.weight"] = (H, H)
            shapes[p + "node_mlp.2.bias"] = (H,)
            if attention:
                shapes[p + "att_mlp.0.weight"] = (1, H)
                shapes[p + "att_mlp.0.bias"] = (1,)
        p = f"egnn.e_block_{i}.gcl_equiv."
        shapes[p + "coord_mlp.0.weight"] = (H, 2 * H + 2)
        shapes[p + "coord_mlp.0.bias"] = (H,)
        shapes[p + "coord_mlp.2.weight"] = (H, H)
        shapes[p + "coord_mlp.2.bias"] = (H,)
        shapes[p + "coord_mlp.4.weight"] = (1, H)
    return shapes


def gamma_param_shapes() -> "OrderedDict[str, Tuple[int, ...]]":
    """GammaNetwork parameters (endiffusion/models/noise_model.py:167-172)."""
    return OrderedDict([
        ("l1.weight", (1, 1)), ("l1.bias", (1,)),
        ("l2.weight", (1024, 1)), ("l2.bias", (1024,)),
        ("l3.weight", (1, 1024)), ("l3.bias", (1,)),
        ("gamma_0", (1,)), ("gamma_1", (1,)),
    ])


def _rng_for(name: str, seed: int) -> np.random.Generator:
    digest = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(digest[:8], "little")))


def _uniform(name: str, seed: int, shape, bound: float) -> np.ndarray:
    return _rng_for(name, seed).uniform(-bound, bound, size=shape).astype(np.float32)


def synthetic_dynamics_state_dict(in_node_nf: int, context_node_nf: int, hidden_nf: int, n_layers: int,
                                  inv_sublayers: int, attention: bool = True, seed: int = 0,
                                  coord_gain: float = 0.001) -> "OrderedDict[str, np.ndarray]":
    """nn.Linear-style fan-in-scaled uniform weights for every dynamics tensor.

    `coord_gain` is the xavier gain of coord_mlp.4 (0.001 in the reference,
    endiffusion/models/layers/egnn_new.py:80-81); fixtures also use a x1000 variant so the
    tanh * coords_range path is numerically exercised.
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in dynamics_param_shapes(in_node_nf, context_node_nf, hidden_nf, n_layers,
                                             inv_sublayers, attention).items():
        if name.endswith("coord_mlp.4.weight"):
            fan_out, fan_in = shape
            bound = coord_gain * math.sqrt(6.0 / (fan_in + fan_out))
        elif name.endswith(".weight"):
            bound = 1.0 / math.sqrt(shape[1])
        else:  # bias: bound from the matching weight's fan-in
            wshape = dynamics_param_shapes(in_node_nf, context_node_nf, hidden_nf, n_layers,
                                           inv_sublayers, attention)[name[:-4] + "weight"]
            bound = 1.0 / math.sqrt(wshape[1])
        out[name] = _uniform(name, seed, shape, bound)
    return out


def gnn_param_shapes(in_node_nf: int, context_node_nf: int, hidden_nf: int, n_layers: int,
                     attention: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered (name -> shape) of EGNN_dynamics_QM9's parameters in mode 'gnn_dynamics' (en_dynamics.py:24-29; GNN, egnn_new.py:208-231:
    node inputs [x | h (incl. time) | context], GCLs without edge attributes, outputs [velocity (3) | h (in_node_nf)])."""
    H = hidden_nf
    fin = in_node_nf + context_node_nf + 3
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    shapes["gnn.embedding.weight"] = (H, fin)
    shapes["gnn.embedding.bias"] = (H,)
    shapes["gnn.embedding_out.weight"] = (3 + in_node_nf, H)
    shapes["gnn.embedding_out.bias"] = (3 + in_node_nf,)
    for i in range(n_layers):
        g = f"gnn.gcl_{i}."
        shapes[g + "edge_mlp.0.weight"] = (H, 2 * H)
        shapes[g + "edge_mlp.0.bias"] = (H,)
        shapes[g + "edge_mlp.2.weight"] = (H, H)
        shapes[g + "edge_mlp.2.bias"] = (H,)
        shapes[g + "node_mlp.0.weight"] = (H, 2 * H)
        shapes[g + "node_mlp.0.bias"] = (H,)
        shapes[g + "node_mlp.2.weight"] = (H, H)
        shapes[g + "node_mlp.2.bias"] = (H,)
        if attention:
            shapes[g + "att_mlp.0.weight"] = (1, H)
            shapes[g + "att_mlp.0.bias"] = (1,)
    return shapes


def synthetic_gnn_state_dict(in_node_nf: int, context_node_nf: int, hidden_nf: int, n_layers: int, attention: bool = False,
                             seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """nn.Linear-style fan-in-scaled uniform weights for every tensor of the 'gnn_dynamics' mode, keyed by name."""
    shapes = gnn_param_shapes(in_node_nf, context_node_nf, hidden_nf, n_layers, attention)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        wshape = shape if name.endswith(".weight") else shapes[name[:-4] + "weight"]
        out[name] = _uniform(name, seed, shape, 1.0 / math.sqrt(wshape[1]))
    return out


def synthetic_gamma_state_dict(seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """PositiveLinear-style init: kaiming-uniform(a=sqrt(5)) - 2 (noise_model.py:92-96)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in gamma_param_shapes().items():
        if name == "gamma_0":
            out[name] = np.array([-5.0], dtype=np.float32)
        elif name == "gamma_1":
            out[name] = np.array([10.0], dtype=np.float32)
        elif name.endswith(".weight"):
            bound = 1.0 / math.sqrt(shape[1])
            out[name] = _uniform("gamma." + name, seed, shape, bound) - np.float32(2.0)
        else:
            wshape = gamma_param_shapes()[name[:-4] + "weight"]
            bound = 1.0 / math.sqrt(wshape[1])
            out[name] = _uniform("gamma." + name, seed, shape, bound)
    return out


def synthetic_state_dict(in_node_nf: int, context_node_nf: int, hidden_nf: int, n_layers: int,
                         inv_sublayers: int = 2, attention: bool = True, seed: int = 0,
                         coord_gain: float = 0.001, pocket: bool = False) -> "OrderedDict[str, np.ndarray]":
    """Full DiffusionQM9 state_dict: `dynamics.*`, `gamma.*` and the `buffer` placeholder
    (endiffusion/train_module/diffusion_qm9.py:95,72,105); with `pocket` also the residue embedding
    `pocket_embed.weight` [21, in_node_nf - 1] (:55-56, N(0,1) like nn.Embedding)."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    if pocket:
        sd["pocket_embed.weight"] = _rng_for("pocket_embed.weight", seed).standard_normal(
            (21, in_node_nf - 1)).astype(np.float32)
    for k, v in synthetic_gamma_state_dict(seed).items():
        sd["gamma." + k] = v
    for k, v in synthetic_dynamics_state_dict(in_node_nf, context_node_nf, hidden_nf, n_layers,
                                              inv_sublayers, attention, seed, coord_gain).items():
        sd["dynamics." + k] = v
    sd["buffer"] = np.zeros(1, dtype=np.float32)
    return sd


def flatten_dynamics(sd: Dict[str, np.ndarray], in_node_nf: int, context_node_nf: int, hidden_nf: int,
                     n_layers: int, inv_sublayers: int, attention: bool = True,
                     prefix: str = "") -> np.ndarray:
    """Concatenate the dynamics tensors in canonical (registration) order into one fp32 blob.

    This canonical blob is what `hd_set_weights` (include/hierdiff_hip.h) consumes and what is
    broadcast between ranks; the library does the kernel-specific repacking itself.
    """
    parts: List[np.ndarray] = []
    for name, shape in dynamics_param_shapes(in_node_nf, context_node_nf, hidden_nf, n_layers,
                                             inv_sublayers, attention).items():
        a = np.asarray(sd[prefix + name], dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{prefix + name}: expected shape {shape}, got {tuple(a.shape)}")
        parts.append(a.reshape(-1))
    return np.concatenate(parts)


def dynamics_param_count(in_node_nf: int, context_node_nf: int, hidden_nf: int, n_layers: int,
                         inv_sublayers: int, attention: bool = True) -> int:
    return int(sum(int(np.prod(s)) for s in dynamics_param_shapes(
        in_node_nf, context_node_nf, hidden_nf, n_layers, inv_sublayers, attention).values()))
