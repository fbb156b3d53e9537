"""hierdiff_amd -- MI355X-native implementation of HierDiff's coarse-grained diffusion sampling hot path.

Public surface mirrors the reference (qiangbo1222/HierDiff, endiffusion/):
  EGNN_dynamics_QM9        models/module/en_dynamics.py
  DiffusionQM9             train_module/diffusion_qm9.py (sample / sample_batches, compute_loss / forward, training_step,
                           validation / test hooks, configure_optimizers)
  EnVariationalDiffusion   equivariant_diffusion/en_diffusion.py sample() signature
  GammaNetwork, PredefinedNoiseSchedule, DistributionNodes
and, of the second stage (top-level models/ of the reference): stage2.E_GCL (models/egnn/gcl.py) and
edge_denoise.Edge_denoise (models/edge_denoise.py: forward VALUES and sample_AR - inference / evaluation only: no backward
through the stage-2 layers, `forward` raises in training mode with autograd recording).
The compute lives in lib/libhierdiff_hip.so (include/hierdiff_hip.h); build it with
`python -m hierdiff_amd.build`.
"""
import os as _os

# dmabuf IPC: the host driver of the MI355X boxes supports no legacy IPC handles, and RCCL between the per-GPU processes (the
# weight broadcast of sharding.py, the gradient all-reduce of trainer.py) fails with `hipIpcGetMemHandle: invalid argument`
# without this.  Set here - importing the package precedes the first HIP call of a rank - so that a rank started by ANY launcher
# (torchrun, Lightning, mpirun, a user's own script) has it, not only the ones bench.py starts itself.  A caller's value wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .diffusion import AttrDict, DiffusionQM9, EnVariationalDiffusion, default_config  # noqa: F401,E402
from .distributions import DistributionNodes  # noqa: F401,E402
from .dynamics import EGNN_dynamics_QM9, Topology, release_cached_memory  # noqa: F401,E402
from .noise_model import GammaNetwork, PredefinedNoiseSchedule  # noqa: F401,E402

__all__ = ["EGNN_dynamics_QM9", "DiffusionQM9", "EnVariationalDiffusion", "GammaNetwork",
           "PredefinedNoiseSchedule", "DistributionNodes", "Topology", "AttrDict", "default_config",
           "release_cached_memory"]
