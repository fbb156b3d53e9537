"""hierdiff_amd -- MI355X-native implementation of HierDiff's coarse-grained diffusion sampling hot path.

Public surface mirrors the reference (qiangbo1222/HierDiff, endiffusion/):
  EGNN_dynamics_QM9        models/module/en_dynamics.py
  DiffusionQM9             train_module/diffusion_qm9.py (sampling half)
  EnVariationalDiffusion   equivariant_diffusion/en_diffusion.py sample() signature
  GammaNetwork, PredefinedNoiseSchedule, DistributionNodes
and, of the second stage (top-level models/ of the reference): stage2.E_GCL (models/egnn/gcl.py) and
edge_denoise.Edge_denoise (models/edge_denoise.py: forward VALUES and sample_AR - inference / evaluation only: no backward
through the stage-2 layers, `forward` raises in training mode with autograd recording).
The compute lives in lib/libhierdiff_hip.so (include/hierdiff_hip.h); build it with
`python -m hierdiff_amd.build`.
"""
from .concurrent import TwoStreamSampler  # noqa: F401
from .diffusion import AttrDict, DiffusionQM9, EnVariationalDiffusion, default_config  # noqa: F401
from .distributions import DistributionNodes  # noqa: F401
from .dynamics import EGNN_dynamics_QM9, Topology, release_cached_memory  # noqa: F401
from .noise_model import GammaNetwork, PredefinedNoiseSchedule  # noqa: F401

__all__ = ["EGNN_dynamics_QM9", "DiffusionQM9", "EnVariationalDiffusion", "GammaNetwork",
           "PredefinedNoiseSchedule", "DistributionNodes", "Topology", "AttrDict", "default_config", "TwoStreamSampler",
           "release_cached_memory"]
