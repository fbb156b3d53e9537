"""Categorical sampler of the number of coarse nodes per molecule.

Behaviour of `DistributionNodes.sample` in endiffusion/models/distributions.py:62-87: a categorical
over the histogram's keys in insertion order, drawn with torch's global RNG on the CPU.
"""
from __future__ import annotations

from typing import Dict, List

import torch
from torch.distributions.categorical import Categorical


class DistributionNodes(torch.nn.Module):
    def __init__(self, histogram: Dict[int, int]):
        super().__init__()
        self.n_nodes: List[int] = list(histogram.keys())
        self.keys = {n: i for i, n in enumerate(self.n_nodes)}
        counts = torch.tensor([float(histogram[n]) for n in self.n_nodes], dtype=torch.float64)
        prob = counts / counts.sum()
        self.prob = prob.float()
        self.m = Categorical(prob)

    @torch.no_grad()
    def sample(self, n_samples: int = 1) -> List[int]:
        return [self.n_nodes[i] for i in self.m.sample((n_samples,)).tolist()]

    def log_prob(self, batch_n_nodes: torch.Tensor) -> torch.Tensor:
        # The reference indexes the probability vector with the node COUNTS themselves (models/distributions.py:94-101:
        # `log_p[batch_n_nodes]`), not with their categorical index `self.keys[n]`; kept as is for drop-in parity
        # (the sampler never calls it).
        assert batch_n_nodes.dim() == 1
        return torch.log(self.prob + 1e-30)[batch_n_nodes]
