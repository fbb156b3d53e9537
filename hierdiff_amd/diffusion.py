"""The reference's `DiffusionQM9` (endiffusion/train_module/diffusion_qm9.py) on MI355X: sampling, loss / NLL, training hooks.

Same entry points and result format as the reference:
  DiffusionQM9.sample(num_samples, device, context=None, pocket_cond=None)        :347-395
  DiffusionQM9.sample_batches(batch_size, num_batches, device, context_range, ..) :397-436
  DiffusionQM9.sample_p_zs_given_zt / sample_p_xh_given_z0 / phi / sigma / alpha   :135-158, 294-345
plus the EDM-style signature named by the north star,
  EnVariationalDiffusion.sample(n_samples, n_nodes, node_mask, edge_mask, context, fix_noise=False)
  (endiffusion/equivariant_diffusion/en_diffusion.py:634-667; dead code in the reference).

The 1000-step loop runs inside libhierdiff_hip.so (hd_sample_loop): about 45 kernels per step at the headline batch
(DESIGN.md section 5), no host sync, by default replayed from one captured hipGraph that is cached per
topology.  The loss / NLL value of the training half is available too (compute_loss, nll, forward).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .distributions import DistributionNodes
from .dynamics import EGNN_dynamics_QM9, Topology, _ptr, _stream
from .geom_stats import GEOM_FRAGMENT_HISTOGRAM
from .noise_model import (GammaNetwork, PredefinedNoiseSchedule, decode_coefficients, evaluate_gamma,
                          schedule_tables, sigma_and_alpha_t_given_s, step_coefficients)

try:  # the reference class is a LightningModule; use it when the package exists so the module nests
    import pytorch_lightning as _pl  # type: ignore
    _Base = _pl.LightningModule
except Exception:  # pragma: no cover - not installed in this image
    _Base = nn.Module


class AttrDict(dict):
    """dict with attribute access, standing in for the OmegaConf node the reference passes around."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_config(hidden_nf: int = 256, n_layers: int = 6, context_node_nf: int = 0, timesteps: int = 1000,
                   inv_sublayers: int = 2, normalization_factor: float = 10.0) -> AttrDict:
    """The production hyper-parameters of endiffusion/conf/model/ddpmgblur.yaml:2-38."""
    return AttrDict(
        pocket=False, node_coarse_type="prop", loss_type="vlb", hcontinous=True, noise_schedule="learned",
        timesteps=timesteps, norm_values=[1.0, 1.0, 1.0], norm_biases=[None, 0.0, 0.0], parametrization="eps",
        include_charges=True, dataset="qm9", conditioning=[], data_augmentation=False,
        pre_noise=AttrDict(noise_schedule="learned", timesteps=timesteps, precision=1e-4),
        dynamics=AttrDict(in_node_nf=0, context_node_nf=context_node_nf, n_dims=3, hidden_nf=hidden_nf,
                          act_fn="silu", n_layers=n_layers, attention=True, condition_time=True, tanh=True,
                          mode="egnn_dynamics", norm_constant=0, inv_sublayers=inv_sublayers, sin_embedding=False,
                          normalization_factor=normalization_factor, aggregation_method="sum"),
        analyze=None,
    )


def _get(cfg, key, default=None):
    try:
        return cfg[key]
    except Exception:
        return getattr(cfg, key, default)


RESIDUE_LIST = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO",
                "SER", "THR", "TRP", "TYR", "VAL"]


def pocket_tensors(protein_data_all):
    """Padded pocket tensors [feat (long, 0 = pad), pos, node_mask, edge_mask] as `sample_batches` builds them
    (diffusion_qm9.py:399-420): residue type index + 1, all-pairs-minus-diagonal edge mask per pocket."""
    feats = [torch.tensor([RESIDUE_LIST.index(r) + 1 for r in p["residue_type"]]) for p in protein_data_all]
    poss = [torch.tensor(np.array(p["coord"])) for p in protein_data_all]
    n, pmax = len(feats), max(f.shape[0] for f in feats)
    feat = torch.zeros(n, pmax, dtype=torch.long)
    pos = torch.zeros(n, pmax, 3)
    nmask = torch.zeros(n, pmax, 1, dtype=torch.bool)
    emask = torch.zeros(n, pmax, pmax, dtype=torch.bool)
    for i, (f, p) in enumerate(zip(feats, poss)):
        k = f.shape[0]
        feat[i, :k] = f
        pos[i, :k] = p
        nmask[i, :k, 0] = True
        emask[i, :k, :k] = ~torch.eye(k, dtype=torch.bool)
    return [feat, pos, nmask, emask]


class DiffusionQM9(_Base):
    """Sampler with the reference's constructor contract: `DiffusionQM9(cfg)` where `cfg` carries the
    keys of conf/model/ddpmgblur.yaml (diffusion_qm9.py:37-115)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.pocket = bool(_get(cfg, "pocket", False))
        self.node_coarse_type = _get(cfg, "node_coarse_type")
        if self.node_coarse_type == "prop":
            self.in_node_nf = 8
        elif self.node_coarse_type == "elem":
            self.in_node_nf = 3
        else:
            raise NotImplementedError("node_coarse_type should be prop or elem")
        if self.pocket:
            self.pocket_embed = nn.Embedding(21, self.in_node_nf)       # diffusion_qm9.py:55-56
        dyn = dict(_get(cfg, "dynamics"))
        dyn["in_node_nf"] = self.in_node_nf
        assert _get(cfg, "loss_type") in {'vlb', 'l2'}
        self.loss_type = _get(cfg, "loss_type")
        self.include_charges = _get(cfg, "include_charges")
        assert _get(cfg, "parametrization") == 'eps'
        if _get(cfg, "noise_schedule") == 'learned':
            assert self.loss_type == 'vlb', 'A noise schedule can only be learned with a vlb objective.'
            self.gamma = GammaNetwork()
        else:
            self.gamma = PredefinedNoiseSchedule(**dict(_get(cfg, "pre_noise")))
        self.hcontinous = _get(cfg, "hcontinous")
        if dyn.get("condition_time", True):
            dyn["in_node_nf"] += 1
        self.dynamics = EGNN_dynamics_QM9(**dyn)
        self.n_dims = dyn["n_dims"]
        self.T = int(_get(cfg, "timesteps"))
        self.parametrization = _get(cfg, "parametrization")
        self.norm_values = list(_get(cfg, "norm_values"))
        self.norm_biases = list(_get(cfg, "norm_biases"))
        # data scaling (diffusion_qm9.py:103-104, 165-179; production: [1,1,1] / [None,0,0], ddpmgblur.yaml:10-11).  The library's
        # decode kernel returns normalised x / h; `_final_decode` applies `unnormalize` when the values are not the unit ones.
        self._unit_norm = [float(v) for v in self.norm_values] == [1.0, 1.0, 1.0] and \
            all(float(b or 0.0) == 0.0 for b in self.norm_biases)
        self.register_buffer('buffer', torch.zeros(1))
        if _get(cfg, "noise_schedule") != 'learned':
            self.check_issues_norm_values()
        self.data_augmentation = _get(cfg, "data_augmentation", False)
        analyze = _get(cfg, "analyze", None)
        if isinstance(analyze, dict):
            histogram = analyze
        elif isinstance(analyze, str):
            import yaml
            with open(analyze) as fh:
                histogram = yaml.safe_load(fh)
        else:
            histogram = GEOM_FRAGMENT_HISTOGRAM
        self.nodes_dist = DistributionNodes(histogram=histogram)
        # sampling knobs of this implementation (not in the reference)
        self.noise_mode = "philox"      # "philox": in-kernel counter RNG; "torch": torch.randn draws
        self.seed = 2022
        self.use_graph = True
        # sample_batches: molecules of consecutive batches run as ONE device batch of at most `merge_batches` molecules and
        # `merge_edges` directed edges (merge_batches = 0: one device batch per call of the reference's loop).  Samples are
        # bit-identical either way (a sample depends on its global id only).  900,000 edges are four headline batches
        # (256 x 30 x 29 each): measured on 8 x 256 GEOM-sized molecules (scratch/geom_job_time.py, fp32) the loop runs at
        # 120.5 molecules/s, device batches of 225 k / 450 k / 900 k edges at 142.5 / 148.8 / 159.2.
        self.merge_batches = 4096
        self.merge_edges = 900_000
        self.debug_checks = False       # True re-enables the reference's host-synchronising asserts
        #: training-mode loss around the network call as two fused launches per direction (csrc/k_loss.hpp) instead of ~350 torch
        #: launches; False = the torch-op path (same arithmetic; what evaluation, pocket models and the CPU run)
        self.fused_loss = True
        self.schedule_gammas = None     # optional [T+1] gamma grid overriding the network (replay a run)
        # "fp64" (default): the schedule network is evaluated once in float64 on the host and rounded - the same table on
        # every machine.  "fp32": evaluated like the reference (float32, a [B,1] column per grid value, CPU BLAS): agrees
        # run for run with a CPU reference on the same host, host-dependent otherwise.
        self.schedule_eval = "fp64"
        self._sched_key = None
        self._sched = None

    def check_issues_norm_values(self, num_stdevs=8):
        """diffusion_qm9.py:117-131 (predefined schedules only)."""
        sigma_0 = float(torch.sqrt(torch.sigmoid(self.gamma(torch.zeros((1, 1))))).reshape(-1)[0])
        max_norm_value = max(self.norm_values[1], self.norm_values[2])
        if sigma_0 * num_stdevs > 1. / max_norm_value:
            raise ValueError(f'Value for normalization value {max_norm_value} probably too large with sigma_0 '
                             f'{sigma_0:.5f} and 1 / norm_value = {1. / max_norm_value}')

    # ------------------------------------------------------------------ schedule algebra (reference API)
    def phi(self, x, t, node_mask, edge_mask, context, mol_shape=None):
        """diffusion_qm9.py:135-138.  Differentiable when autograd is recording (see EGNN_dynamics_QM9._forward)."""
        return self.dynamics._forward(t, x, node_mask, edge_mask, context, mol_shape)

    def inflate_batch_array(self, array, target):
        return array.view((array.size(0),) + (1,) * (len(target.size()) - 1))

    def sigma(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(gamma)), target_tensor)

    def alpha(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(-gamma)), target_tensor)

    def SNR(self, gamma):
        return torch.exp(-gamma)

    def sigma_and_alpha_t_given_s(self, gamma_t, gamma_s, target_tensor):
        s2, s, a = sigma_and_alpha_t_given_s(gamma_t, gamma_s)
        return (self.inflate_batch_array(s2, target_tensor), self.inflate_batch_array(s, target_tensor),
                self.inflate_batch_array(a, target_tensor))

    def compute_x_pred(self, net_out, zt, gamma_t):
        sigma_t = self.sigma(gamma_t, target_tensor=net_out)
        alpha_t = self.alpha(gamma_t, target_tensor=net_out)
        return 1. / alpha_t * (zt - sigma_t * net_out)

    def unnormalize(self, x, h, node_mask):
        x = x * self.norm_values[0]
        h = (h * self.norm_values[1] + self.norm_biases[1]) * node_mask
        return x, h

    # ------------------------------------------------------------------ loss / NLL, forward value (reference API)
    # diffusion_qm9.py:160-172, 206-292, 460-751.  The network calls go through the HIP dynamics (per-row t); the
    # few element-wise terms around them are torch ops on the same device.  Under torch.no_grad() they return values
    # (validation NLL); with autograd recording they are differentiable (training_step): see `phi`.
    def subspace_dimensionality(self, node_mask):
        return (torch.sum(node_mask.squeeze(2), dim=1) - 1) * self.n_dims

    def normalize(self, x, h, node_mask):
        x = x / self.norm_values[0]
        delta_log_px = -self.subspace_dimensionality(node_mask) * math.log(self.norm_values[0])
        h = (h - self.norm_biases[1]) / self.norm_values[1] * node_mask
        return x, h, delta_log_px

    def _gamma_rows(self, t, key, gammas):
        """gamma at the [B,1] times `t`: fp64 evaluation rounded once (noise_model.evaluate_gamma) unless the caller
        replays recorded values (`gammas[key]`).  Every time the loss asks for lies on the grid k / T, k = -1 .. T
        (s = (t_int - 1) / T, t = t_int / T, 0, 1: diffusion_qm9.py:541-552), so without autograd the values come from a
        device-resident table of the T + 2 grid values - the same fp64 evaluation at the same fp32 arguments, tabulated
        once per version of the schedule parameters - and a loss evaluation needs no host round trip."""
        if gammas is not None and key in gammas:
            return torch.as_tensor(gammas[key], dtype=torch.float32, device=t.device).view(-1, 1)
        # training a LEARNED schedule: the network is part of the autograd graph (fp32, like the reference).  A predefined
        # schedule has no trainable parameter, and an evaluation call needs no graph: both read the table, so that the same
        # batch gives the same NLL whatever the grad mode
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.gamma.parameters()):
            return self.gamma(t).view(-1, 1)
        gparams = list(self.gamma.state_dict(keep_vars=True).values())
        ver = (self.T, str(t.device), _lib.optimizer_generation()) + tuple((p.data_ptr(), p._version) for p in gparams)
        guard = self.__dict__.setdefault("_gamma_grid_guard", _lib.ImageGuard())
        if not guard.valid(ver, gparams):             # key AND content of the schedule parameters (_lib.ImageGuard)
            k = torch.arange(-1, self.T + 1, dtype=torch.float32).view(-1, 1)
            self._gamma_grid = evaluate_gamma(self.gamma, k / self.T).view(-1).to(t.device)
            guard.store(ver, gparams)
        idx = torch.round(t.to(torch.float32) * self.T).long().view(-1) + 1
        return self._gamma_grid[idx].view(-1, 1)

    def compute_error(self, net_out, gamma_t, eps):
        err = (eps - net_out) ** 2
        err = err.reshape(err.size(0), -1).sum(-1)
        if self.training and self.loss_type == 'l2':
            err = err / ((self.n_dims + self.in_node_nf) * net_out.shape[1])
        return err

    def kl_prior(self, xh, node_mask, gamma_T=None):
        B = xh.size(0)
        if gamma_T is None:
            gamma_T = self._gamma_rows(torch.ones((B, 1), device=xh.device), "gamma_T", None)
        nm = node_mask.to(xh.dtype)
        mu = self.alpha(gamma_T, xh) * xh
        sig = torch.sqrt(torch.sigmoid(gamma_T)).view(-1)
        sig3 = sig.view(-1, 1, 1)
        kl_h = ((torch.log(1.0 / sig3) + 0.5 * (sig3 ** 2 + mu[:, :, self.n_dims:] ** 2) - 0.5) * nm).reshape(B, -1).sum(-1)
        d = self.subspace_dimensionality(nm)
        mu2 = (mu[:, :, :self.n_dims] ** 2).reshape(B, -1).sum(-1)
        kl_x = d * torch.log(1.0 / sig) + 0.5 * (d * sig ** 2 + mu2) - 0.5 * d
        return kl_x + kl_h

    def log_constants_p_x_given_z0(self, x, node_mask, gamma_0=None):
        n_nodes = node_mask.squeeze(2).sum(1)
        if gamma_0 is None:
            gamma_0 = self._gamma_rows(torch.zeros((x.size(0), 1), device=x.device), "gamma_0", None)
        return (n_nodes - 1) * self.n_dims * (-0.5 * gamma_0.view(-1) - 0.5 * math.log(2 * math.pi))

    def log_constants_p_h_given_z0(self, h, node_mask, gamma_0=None):
        n_nodes = node_mask.squeeze(2).sum(1)
        if gamma_0 is None:
            gamma_0 = self._gamma_rows(torch.zeros((h.size(0), 1), device=h.device), "gamma_0", None)
        return n_nodes * self.in_node_nf * (-0.5 * gamma_0.view(-1) - 0.5 * math.log(2 * math.pi))

    def log_pxh_given_z0_without_constants(self, x, h, z_t, gamma_0, eps, net_out, node_mask, epsilon=1e-10):
        int_nf, cont_nf = (5, 3) if self.node_coarse_type == 'prop' else (3, 0)
        nd = self.n_dims
        z_h_int = z_t[:, :, nd:nd + int_nf]
        log_px = -0.5 * self.compute_error(net_out[:, :, :nd], gamma_0, eps[:, :, :nd])
        # the reference slices the continuous-feature prediction with a stride (`[: nd+int_nf : nd+int_nf+cont_nf]`,
        # diffusion_qm9.py:477), i.e. column 0 broadcast against the noise columns; kept for drop-in parity
        log_ph = -0.5 * self.compute_error(net_out[:, :, :nd + int_nf:nd + int_nf + cont_nf], gamma_0,
                                           eps[:, :, nd + int_nf:nd + int_nf + cont_nf])
        sigma_0_int = self.sigma(gamma_0, target_tensor=z_t) * self.norm_values[2]
        h_integer = torch.round(h[:, :, :int_nf] * self.norm_values[2] + self.norm_biases[2]).long()
        centred = h_integer - (z_h_int * self.norm_values[2] + self.norm_biases[2])
        cdf = lambda v: 0.5 * (1. + torch.erf(v / math.sqrt(2)))
        log_int = torch.log(cdf((centred + 0.5) / sigma_0_int) - cdf((centred - 0.5) / sigma_0_int) + epsilon)
        log_int = (log_int * node_mask).reshape(x.size(0), -1).sum(-1)
        return log_px + log_ph + log_int

    def compute_loss(self, x, h, node_mask, edge_mask, context, t0_always, mol_shape=None,
                     t_int=None, eps=None, eps0=None, gammas=None):
        """Forward value of the variational bound estimator / simple loss (diffusion_qm9.py:530-673).  Nodes behind
        `mol_shape` (pocket residues) are fixed: they enter the network un-noised and the loss covers the first
        mol_shape nodes (:553-579).  `t_int` [B,1], `eps`, `eps0` [B,mol,3+F] replay recorded draws (otherwise
        torch.randint / torch.randn on x.device, in the reference's order); `gammas` replays schedule values (keys
        gamma_s, gamma_t, gamma_0, gamma_T)."""
        B = x.size(0)
        dev = x.device
        mol = x.size(1) if mol_shape is None else int(mol_shape)
        x, x_fix = x[:, :mol], x[:, mol:]
        h, h_fix = h[:, :mol], h[:, mol:]
        node_mask_all = node_mask
        node_mask = node_mask[:, :mol]
        nm = node_mask.to(torch.float32)
        if t_int is None:
            t_int = torch.randint(1 if t0_always else 0, self.T + 1, size=(B, 1), device=dev).float()
        t_int = torch.as_tensor(t_int, dtype=torch.float32, device=dev).view(B, 1)
        s, t = (t_int - 1) / self.T, t_int / self.T
        t_is_zero = (t_int == 0).float().view(-1)
        if (gammas is None and x.is_cuda and torch.is_grad_enabled() and self.training
                and any(p.requires_grad for p in self.gamma.parameters())):
            # training a learned schedule: the four schedule values of the loss (diffusion_qm9.py:541-552) from ONE pass of the
            # network over [s; t; 0; 1] instead of four (each ~125 small launches forward + backward)
            gamma_s, gamma_t, gamma_0, gamma_T = self.gamma(torch.cat([s, t, torch.zeros_like(t), torch.ones_like(t)], dim=0)).view(4, B, 1)
        else:
            gamma_s, gamma_t = self._gamma_rows(s, "gamma_s", gammas), self._gamma_rows(t, "gamma_t", gammas)
            gamma_0 = self._gamma_rows(torch.zeros_like(t), "gamma_0", gammas)
            gamma_T = self._gamma_rows(torch.ones_like(t), "gamma_T", gammas)
        if eps is None:
            eps = self.sample_combined_position_feature_noise(B, mol, node_mask)
        eps = torch.as_tensor(eps, dtype=torch.float32, device=dev)
        xh = torch.cat([x, h], dim=2).to(torch.float32)
        if (self.fused_loss and not t0_always and mol_shape is None and x.is_cuda and torch.is_grad_enabled() and self.training
                and xh.shape[2] == self.n_dims + self.in_node_nf):
            # the training loss around the network call as one launch per direction (training.vlb_zt / vlb_loss, csrc/k_loss.hpp);
            # everything this branch skips below is the same arithmetic in ~350 element-wise launches, kept for every other case
            # (evaluation, pocket models, CPU) and as the oracle of tests/test_gpu_training.py
            from .training import vlb_loss, vlb_zt
            self._check_mean_zero(x, node_mask)
            gam = torch.stack([gamma_s.reshape(B), gamma_t.reshape(B), gamma_0.reshape(B), gamma_T.reshape(B)])
            xh, eps = xh.contiguous(), eps.contiguous()
            z_t = vlb_zt(xh, eps, gam[1])
            net_out = self.phi(z_t, t, node_mask_all, edge_mask, context, mol_shape=mol)
            int_nf, cont_nf = (5, 3) if self.node_coarse_type == 'prop' else (3, 0)
            l2_train = self.loss_type == 'l2'
            consts = (int_nf, cont_nf, l2_train, float(self.T), float(self.norm_values[2]), float(self.norm_biases[2]), 0.0)
            loss, error = vlb_loss(net_out, z_t, gam, xh, eps, nm.reshape(B, mol).contiguous(), t_int.reshape(B).contiguous(), consts)
            return loss, {'t': t_int.squeeze(), 'loss_t': loss.squeeze(), 'error': error.squeeze()}
        xh_fix = torch.cat([x_fix, h_fix], dim=2).to(torch.float32)
        self._check_mean_zero(x, node_mask)
        z_t = self.alpha(gamma_t, x) * xh + self.sigma(gamma_t, x) * eps
        self._check_mean_zero(z_t[:, :, :self.n_dims], node_mask)
        z_t = torch.cat([z_t, xh_fix], dim=1)
        net_out = self.phi(z_t, t, node_mask_all, edge_mask, context, mol_shape=mol)[:, :mol]
        error = self.compute_error(net_out, gamma_t, eps)
        l2_train = self.training and self.loss_type == 'l2'
        snr_weight = torch.ones_like(error) if l2_train else (self.SNR(gamma_s - gamma_t) - 1).view(-1)
        loss_t_larger_than_zero = 0.5 * snr_weight * error
        neg_log_constants = -self.log_constants_p_x_given_z0(x, nm, gamma_0) - self.log_constants_p_h_given_z0(h, nm, gamma_0)
        if l2_train:
            neg_log_constants = torch.zeros_like(neg_log_constants)
        kl_prior = self.kl_prior(xh, nm, gamma_T)
        if t0_always:
            if eps0 is None:
                eps0 = self.sample_combined_position_feature_noise(B, mol, node_mask)
            eps0 = torch.as_tensor(eps0, dtype=torch.float32, device=dev)
            z_0 = torch.cat([self.alpha(gamma_0, x) * xh + self.sigma(gamma_0, x) * eps0, xh_fix], dim=1)
            net0 = self.phi(z_0, torch.zeros_like(t), node_mask_all, edge_mask, context, mol_shape=mol)[:, :mol]
            loss_term_0 = -self.log_pxh_given_z0_without_constants(x, h, z_0[:, :mol], gamma_0, eps0, net0, nm)
            loss = kl_prior + self.T * loss_t_larger_than_zero + neg_log_constants + loss_term_0
        else:
            loss_term_0 = -self.log_pxh_given_z0_without_constants(x, h, z_t[:, :mol], gamma_t, eps, net_out, nm)
            loss_t = loss_term_0 * t_is_zero + (1 - t_is_zero) * loss_t_larger_than_zero
            estimator = loss_t if l2_train else (self.T + 1) * loss_t
            loss = kl_prior + estimator + neg_log_constants
        return loss, {'t': t_int.squeeze(), 'loss_t': loss.squeeze(), 'error': error.squeeze()}

    def nll(self, x, h, node_mask=None, edge_mask=None, context=None, mol_shape=None, **replay):
        """Loss if training (value only), NLL estimate if eval (diffusion_qm9.py:675-699)."""
        x, h, delta_log_px = self.normalize(x, h, node_mask.to(torch.float32))
        if self.training and self.loss_type == 'l2':
            delta_log_px = torch.zeros_like(delta_log_px)
        loss, _ = self.compute_loss(x, h, node_mask, edge_mask, context, t0_always=not self.training,
                                    mol_shape=mol_shape, **replay)
        return loss - delta_log_px

    def forward(self, batch, **replay):
        """`{"loss": mean NLL}` for a reference data batch (keys positions, atom_mask, edge_mask, node_feature; with a
        context model, context; with a pocket model, protein_pos, protein_feat, protein_feat_mask,
        protein_edge_mask) - diffusion_qm9.py:701-751."""
        x, node_mask, edge_mask, h = batch['positions'], batch['atom_mask'], batch['edge_mask'], batch["node_feature"]
        mol_shape = None
        if self.pocket:
            mol_shape = x.shape[1]
            x = torch.cat([x, batch["protein_pos"].to(x.dtype)], dim=1)
            node_mask = torch.cat([node_mask, batch["protein_feat_mask"]], dim=1)
            P = batch["protein_edge_mask"].shape[1]
            em = torch.zeros(edge_mask.shape[0], mol_shape + P, mol_shape + P, dtype=edge_mask.dtype, device=edge_mask.device)
            em[:, :mol_shape, :mol_shape] = edge_mask.view(-1, mol_shape, mol_shape)
            em[:, mol_shape:, mol_shape:] = batch["protein_edge_mask"]
            edge_mask = em
            h = torch.cat([h, self.pocket_embed(batch["protein_feat"]).to(h.dtype)], dim=1)
        nm = node_mask.to(x.dtype)
        if self.debug_checks:                       # models/utils.py:47-50
            bad = (x * (1 - nm)).abs().sum().item()
            assert bad < 1e-5, f'Error {bad} too high'
        fix = x.size(1) if mol_shape is None else mol_shape
        # remove_mean_with_mask(x, node_mask, fix_size=mol_shape): the mean of the first mol_shape nodes is taken off
        # every valid node, pocket residues included (models/utils.py:51-56)
        x = x - (x[:, :fix].sum(1, keepdim=True) / nm[:, :fix].sum(1, keepdim=True)) * nm
        context = batch['context'] if self.dynamics.context_node_nf > 0 else None
        bs, n_nodes, _ = x.size()
        edge_mask = edge_mask.reshape(bs, n_nodes * n_nodes)
        self._check_masked(x, node_mask, "assert_correctly_masked")
        neg_log_pxh = self.nll(x, h, node_mask, edge_mask, context=context, mol_shape=mol_shape, **replay)
        return {"loss": neg_log_pxh.mean(0)}

    def stage_batch(self, batch, device=None):
        """A collated HOST batch (the dict a DataLoader yields) -> the device batch `forward` / `training_step` take, without
        a host wait: tensors travel through pinned copies in stream order, and the masks' topology is laid out from the
        host copies on the way (`EGNN_dynamics_QM9.stage_masks`) - so a loop that stages batch k+1 after launching step k
        overlaps the layout with the GPU's work.  This is the place of Lightning's transfer_batch_to_device in the reference's
        trainer.  Pocket batches (their masks are composed on the device in `forward`) and tensors already on the device
        are moved as they are."""
        from .dynamics import _to_device_async
        dev = self.dynamics._device() if device is None else torch.device(device)
        out = {k: (_to_device_async(v, dev) if torch.is_tensor(v) else v) for k, v in batch.items()
               if k not in ("atom_mask", "edge_mask")}
        nm, em = batch.get("atom_mask"), batch.get("edge_mask")
        if (not self.pocket and torch.is_tensor(nm) and nm.device.type == "cpu"
                and (em is None or (torch.is_tensor(em) and em.device.type == "cpu"))):
            nm_d, em_d = self.dynamics.stage_masks(nm, em, dev)
            out["atom_mask"] = nm_d
            if "edge_mask" in batch:
                out["edge_mask"] = em_d
        else:
            for k in ("atom_mask", "edge_mask"):
                if k in batch:
                    out[k] = _to_device_async(batch[k], dev) if torch.is_tensor(batch[k]) else batch[k]
        return out

    def training_step(self, batch, batch_idx=0):
        """diffusion_qm9.py:774-777: differentiable mean loss of the batch (call .backward() on it, then - multi-GPU -
        hierdiff_amd.sharding.allreduce_gradients, the DDP step of conf/trainer/default.yaml:2-3)."""
        loss = self.forward(batch)["loss"]
        if hasattr(self, "log") and _Base is not nn.Module:
            self.log("train_loss", loss, on_epoch=True, prog_bar=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        """diffusion_qm9.py:779-781 (value only)."""
        return self.forward(batch)

    test_step = validation_step

    # ------------------------------------------------------------------ epoch-end hooks and optimiser (Lightning side of the module)
    # diffusion_qm9.py:753-801, 871-879.  With pytorch_lightning installed the base class provides `log`, `all_gather`,
    # `global_rank`; without it (this image) the same hooks work over torch.distributed, or on one process.
    def _rank(self) -> int:
        if _Base is not nn.Module:
            return int(self.global_rank)
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def _gather_ranks(self, value: torch.Tensor) -> torch.Tensor:
        """[world, ...] stack of `value` from every rank (LightningModule.all_gather's shape; [1, ...] on one process).  Every
        rank must hold the same shape, as under Lightning."""
        if _Base is not nn.Module:
            out = self.all_gather(value)
            return out if out.dim() > value.dim() else out.unsqueeze(0)
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return value.unsqueeze(0)
        parts = [torch.empty_like(value) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, value.contiguous())
        return torch.stack(parts)

    def _gather_result(self, result):
        """List of per-step dicts -> one dict of tensors: first the steps are joined (tensors concatenated, scalars stacked into
        a vector), then the ranks (diffusion_qm9.py:753-766)."""
        keys = list(result[0].keys())
        steps = {}
        for key in keys:
            first = result[0][key]
            if first.dim() > 0:
                steps[key] = torch.cat([r[key] for r in result])
            else:
                steps[key] = torch.stack([r[key].detach() for r in result]).to(first)
        return {key: torch.cat(list(self._gather_ranks(steps[key]))) for key in keys}

    def _compute_metrics(self, result):
        """diffusion_qm9.py:768-772: the epoch metric is the mean of the gathered per-step losses."""
        return {'loss': result['loss'].mean()}

    def _log(self, name, value, **kw):
        if _Base is not nn.Module:
            self.log(name, value, **kw)
        else:                                   # no Lightning: the last logged values stay readable on the module
            self.logged = getattr(self, "logged", {})
            self.logged[name] = value

    def validation_epoch_end(self, result):
        """diffusion_qm9.py:787-795."""
        metrics = self._compute_metrics(self._gather_result(result))
        self._log("val_loss", metrics["loss"], on_epoch=True, prog_bar=True)

    def test_epoch_end(self, result):
        """diffusion_qm9.py:797-800 (logged by rank 0 only)."""
        metrics = self._compute_metrics(self._gather_result(result))
        if self._rank() == 0:
            self._log("test/ppl", metrics["loss"], on_epoch=True)

    def configure_optimizers(self):
        """diffusion_qm9.py:871-879: `[optimizer], [scheduler]` from `cfg.optim` / `cfg.scheduler` (hydra nodes with a `_target_`,
        conf/optim/adamw.yaml, conf/scheduler/step.yaml).  hydra's `instantiate` is used when the package exists; otherwise the
        `_target_` is resolved here - only names under `torch.optim` are accepted - and a cfg without these nodes gets the
        reference's shipped values (AdamW 4e-4 / 4e-8, StepLR 15 / 0.1: hierdiff_amd.trainer.configure_optimizers, fused update
        on the GPU).  A scheduler node that asks for `num_training_steps` needs the Lightning trainer and is not supported
        without it."""
        from .trainer import configure_optimizers as _defaults
        optim_cfg, sched_cfg = _get(self.cfg, "optim", None), _get(self.cfg, "scheduler", None)
        if optim_cfg is None:
            opt, sched = _defaults(self)
            if sched_cfg is not None:
                sched = self._instantiate(sched_cfg, opt)
            return [opt], [sched]
        opt = self._instantiate(optim_cfg, self.parameters())
        sched = self._instantiate(sched_cfg, opt) if sched_cfg is not None else torch.optim.lr_scheduler.StepLR(opt, step_size=15, gamma=0.1)
        return [opt], [sched]

    @staticmethod
    def _instantiate(node, *args):
        node = dict(node)
        if "num_training_steps" in node:
            raise NotImplementedError("schedulers keyed on num_training_steps need the Lightning trainer (diffusion_qm9.py:804-869)")
        try:
            from hydra.utils import instantiate  # type: ignore
            return instantiate(node, *args)
        except ImportError:
            pass
        target = str(node.pop("_target_"))
        if not target.startswith("torch.optim."):
            raise ValueError(f"_target_ {target!r}: only torch.optim.* optimisers / lr_schedulers are resolved without hydra")
        obj = torch.optim
        for part in target.split(".")[2:]:
            obj = getattr(obj, part)
        return obj(*args, **node)

    # ------------------------------------------------------------------ HIP plumbing
    def _lib_handle(self, synced: bool = False):
        """The dynamics' handle with its weight image confirmed (key + content digest, `sync_weights`).  `synced`: the caller has just
        evaluated the network through this handle (`phi`), which confirmed it - the check costs a stream wait, and the step-by-step
        samplers would pay it twice per diffusion step."""
        if not (synced and getattr(self.dynamics, "mode", "egnn_dynamics") == "egnn_dynamics"):
            self.dynamics.sync_weights()
        return self.dynamics._handle()

    def _schedule(self, rows: int = 1):
        """Tabulated schedule, uploaded to the handle (hd_set_schedule); recomputed when gamma changes."""
        handle = self._lib_handle()          # creates the handle if needed: its generation is part of the key (a new
        # handle - other precision, other device - has no schedule yet, even if it re-uses a freed handle's address)
        key = (self.T, self.dynamics._handle_gen, _lib.optimizer_generation()) + tuple(
            (p.data_ptr(), p._version) for p in self.gamma.parameters()) + (
                id(self.schedule_gammas), self.schedule_eval, rows if self.schedule_eval == "fp32" else 0)
        gparams = list(self.gamma.parameters())
        guard = self.__dict__.setdefault("_sched_guard", _lib.ImageGuard())
        if self._sched_key is None:
            guard.clear()
        if not guard.valid(key, gparams):             # key AND content of the schedule parameters (_lib.ImageGuard)
            tabs = schedule_tables(self.gamma, self.T, self.schedule_gammas, self.schedule_eval, rows)
            tau = tabs["tau"].numpy().astype(np.float32)
            coef = tabs["coef"].numpy().astype(np.float32).reshape(-1)
            _lib.check(_lib.load().hd_set_schedule(
                handle, self.T, tau.ctypes.data_as(C.POINTER(C.c_float)),
                coef.ctypes.data_as(C.POINTER(C.c_float))), "hd_set_schedule")
            self._sched = tabs
            guard.store(key, gparams)
            self._sched_key = key
        return self._sched

    def _check_masked(self, x, node_mask, what):
        if self.debug_checks:
            bad = (x * (~node_mask.bool())).abs().max().item()
            assert bad < 1e-4, f'{what}: variables not masked properly ({bad})'

    def _check_mean_zero(self, x, node_mask):
        if self.debug_checks:
            self._check_masked(x, node_mask, "assert_mean_zero_with_mask")
            largest = x.abs().max().item()
            err = torch.sum(x, dim=1, keepdim=True).abs().max().item()
            assert err / (largest + 1e-10) < 1e-2, f'Mean is not zero, relative_error {err / (largest + 1e-10)}'

    # ------------------------------------------------------------------ single transitions (reference API)
    def sample_combined_position_feature_noise(self, n_samples, n_nodes, node_mask):
        """diffusion_qm9.py:445-456 with torch.randn draws on node_mask.device (x first, then h)."""
        dev = node_mask.device
        nm = node_mask.to(torch.float32)
        raw_x = torch.randn((n_samples, n_nodes, self.n_dims), device=dev)
        raw_h = torch.randn((n_samples, n_nodes, self.in_node_nf), device=dev)
        zx = raw_x * nm
        zx = zx - (zx.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
        return torch.cat([zx, raw_h * nm], dim=2)

    def _combine_raw(self, raw, node_mask, B):
        """z_T from an injected (randn_x [b,N,3], randn_h [b,N,F]) pair: masked, x centred (diffusion_qm9.py:445-456), b = 1 broadcast."""
        dev = node_mask.device
        nm = node_mask.to(torch.float32)
        rx, rh = (r.to(dev, torch.float32) for r in raw)
        zx = rx * nm
        zx = zx - (zx.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
        z = torch.cat([zx, rh * nm], dim=2)
        return z.expand(B, -1, -1).contiguous() if z.shape[0] == 1 and B > 1 else z.contiguous()

    @torch.no_grad()
    def sample_p_zs_given_zt(self, s, t, zt, node_mask, edge_mask, context, fix_noise=False, mol_shape=None,
                             raw_noise: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                             gammas: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """zs ~ p(zs | zt) (diffusion_qm9.py:312-345).  Returns [B, mol_shape, D]: the reference
        appends an empty slice because `zt` was re-bound to its first mol_shape nodes (:326,:345).
        raw_noise / gammas optionally inject the two randn draws / (gamma_s, gamma_t)."""
        dev = zt.device
        B, N, D = zt.shape
        mol = N if mol_shape is None else min(int(mol_shape), N)
        if gammas is None:
            gammas = (evaluate_gamma(self.gamma, s), evaluate_gamma(self.gamma, t))
        coef = step_coefficients(gammas[0].detach().float().cpu().reshape(-1, 1),
                                 gammas[1].detach().float().cpu().reshape(-1, 1)).to(dev)
        zt_c = zt.detach().to(torch.float32).contiguous()
        eps = self.phi(zt_c, t, node_mask, edge_mask, context, mol_shape)
        self._check_mean_zero(zt_c[:, :mol, :self.n_dims], node_mask[:, :mol])
        nb = 1 if fix_noise else B
        if raw_noise is None:
            raw_x = torch.randn((nb, mol, self.n_dims), device=dev)
            raw_h = torch.randn((nb, mol, self.in_node_nf), device=dev)
        else:
            raw_x, raw_h = (r.to(dev, torch.float32).contiguous() for r in raw_noise)
        topo = self.dynamics.topology(node_mask, edge_mask, B, N)
        zs = torch.empty((B, mol, D), device=dev, dtype=torch.float32)
        _lib.check(_lib.load().hd_posterior_step(
            self._lib_handle(synced=True), topo.ptr, zt_c.data_ptr(), eps.data_ptr(), coef.data_ptr(), coef.shape[0],
            raw_x.data_ptr(), raw_h.data_ptr(), raw_x.shape[0], mol, zs.data_ptr(), _stream(dev)), "hd_posterior_step")
        return zs

    @torch.no_grad()
    def sample_p_xh_given_z0(self, z0, node_mask, edge_mask, context, fix_noise=False,
                             raw_noise: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                             gamma_0: Optional[torch.Tensor] = None):
        """x ~ p(x | z0), h = z0 features (diffusion_qm9.py:294-310)."""
        dev = z0.device
        B, N, D = z0.shape
        zeros = torch.zeros(size=(B, 1), device=dev)
        if gamma_0 is None:
            gamma_0 = evaluate_gamma(self.gamma, torch.zeros(1, 1))
        z0_c = z0.detach().to(torch.float32).contiguous()
        eps = self.phi(z0_c, zeros, node_mask, edge_mask, context)
        coef3 = decode_coefficients(gamma_0.detach().float().cpu()).numpy()
        return self._final_decode(z0_c, eps, node_mask, edge_mask, coef3, fix_noise, raw_noise)

    def _final_decode(self, z0, eps, node_mask, edge_mask, coef3, fix_noise, raw_noise, philox=None):
        """raw_noise: (randn_x, randn_h) or None -> torch.randn; philox=(sample_id_base, draw) uses the
        library's counter RNG instead."""
        dev = z0.device
        B, N, D = z0.shape
        nb = 1 if fix_noise else B
        raw_x = raw_h = None
        if philox is None:
            if raw_noise is None:
                raw_x = torch.randn((nb, N, self.n_dims), device=dev)
                raw_h = torch.randn((nb, N, self.in_node_nf), device=dev)
            else:
                raw_x, raw_h = (r.to(dev, torch.float32).contiguous() for r in raw_noise)
            nb = raw_x.shape[0]
        topo = self.dynamics.topology(node_mask, edge_mask, B, N)
        x = torch.empty((B, N, self.n_dims), device=dev, dtype=torch.float32)
        h = torch.empty((B, N, self.in_node_nf), device=dev, dtype=torch.float32)
        c3 = np.ascontiguousarray(coef3, dtype=np.float32)
        _lib.check(_lib.load().hd_final_decode(
            self._lib_handle(synced=True), topo.ptr, z0.data_ptr(), eps.data_ptr(), c3.ctypes.data_as(C.POINTER(C.c_float)),
            _ptr(raw_x), _ptr(raw_h), nb, self.seed, philox[0] if philox else 0, philox[1] if philox else 0,
            int(fix_noise), x.data_ptr(), h.data_ptr(), _stream(dev)), "hd_final_decode")
        if not self._unit_norm:         # `unnormalize` (:174-179); h is already masked: (h nv1 + nb1) mask = h_masked nv1 + nb1 mask
            x = x * float(self.norm_values[0])
            h = h * float(self.norm_values[1]) + float(self.norm_biases[1] or 0.0) * node_mask.reshape(B, N, 1).to(h.dtype)
        return x, h

    def sample_normal(self, mu, sigma, node_mask, fix_noise=False):
        bs = 1 if fix_noise else mu.size(0)
        return mu + sigma * self.sample_combined_position_feature_noise(bs, mu.size(1), node_mask)

    # ------------------------------------------------------------------ full reverse process
    @torch.no_grad()
    def sample_from_masks(self, node_mask: torch.Tensor, edge_mask: Optional[torch.Tensor], context=None,
                          fix_noise: bool = False, raw_noises: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]] = None,
                          sample_id_base: int = 0, z_init: Optional[torch.Tensor] = None, pocket=None):
        """z_T -> (x, h) for given masks: draw z_T, T posterior steps, final decode.

        raw_noises: optional T+2 (randn_x[b,N,3], randn_h[b,N,F]) pairs in the reference's draw order
        (z_T, steps s=T-1..0, decode) for bit-for-bit comparable trajectories; otherwise noise comes
        from `noise_mode` ("philox": sample ids sample_id_base + b, independent of batch split).
        pocket: optional (pos [B,P,3], feat [B,P,F] already embedded, node_mask [B,P,1], edge_mask [B,P,P]) of fixed
        residue nodes: appended to z for every network call with a block-diagonal edge mask and never updated
        (diffusion_qm9.py:362-371,381-382); the final decode sees the molecule alone (:386-387)."""
        dev = node_mask.device
        if dev.type != "cuda":
            raise _lib.HierDiffHipError("sampling runs only on an MI355X (no CPU fallback)")
        B, N = node_mask.shape[0], node_mask.shape[1]
        D = self.n_dims + self.in_node_nf
        lib = _lib.load()
        h = self._lib_handle()
        tabs = self._schedule(rows=B)
        topo = self.dynamics.topology(node_mask, edge_mask, B, N)
        ctx = None
        if self.dynamics.context_node_nf > 0:
            if context is None:
                raise ValueError("context required")
            ctx = context.to(dev, torch.float32).reshape(B * N, -1).contiguous()
        topo_loop, tail = topo, None
        if pocket is not None:
            if ctx is not None or self.noise_mode == "torch":
                raise NotImplementedError("pocket conditioning: context=None and library/injected noise only")
            p_pos, p_feat, p_nm, p_em = pocket
            P = p_pos.shape[1]
            tail = torch.cat([p_pos.to(dev, torch.float32), p_feat.to(dev, torch.float32)], dim=-1)
            nmb = node_mask.to(torch.bool)
            em_mol = edge_mask.to(torch.bool).reshape(B, N, N) if edge_mask is not None else \
                (nmb & nmb.transpose(1, 2) & ~torch.eye(N, dtype=torch.bool, device=dev)[None])
            self._nm_cat = torch.cat([nmb, p_nm.to(dev).bool()], dim=1)
            self._em_cat = torch.zeros(B, N + P, N + P, dtype=torch.bool, device=dev)
            self._em_cat[:, :N, :N] = em_mol
            self._em_cat[:, N:, N:] = p_em.to(dev).bool()
            topo_loop = self.dynamics.topology(self._nm_cat, self._em_cat, B, N + P)
        stream = _stream(dev)

        def run_loop(z_mol, rx, rh, rows, seed, base):
            """T posterior steps on [B,N,D]; with a pocket the fixed rows ride along behind the molecule."""
            zz = z_mol if tail is None else torch.cat([z_mol, tail], dim=1).contiguous()
            _lib.check(lib.hd_sample_loop(h, topo_loop.ptr, zz.data_ptr(), _ptr(ctx), -1 if tail is None else N, T, 0,
                                          _ptr(rx), _ptr(rh), rows, seed, base, int(self.use_graph), stream),
                       "hd_sample_loop")
            return zz if tail is None else zz[:, :N].contiguous()

        nb = 1 if fix_noise else B
        T = self.T
        z = torch.empty((B, N, D), device=dev, dtype=torch.float32)
        gnn = getattr(self.dynamics, "mode", "egnn_dynamics") == "gnn_dynamics"
        if gnn and (pocket is not None or z_init is not None):
            raise NotImplementedError("mode 'gnn_dynamics': plain sampling only (no pocket, no z_init)")
        if gnn and raw_noises is not None:
            # step by step (the fused loop evaluates the egnn network): injected draws in the reference's order
            assert len(raw_noises) == T + 2, "need T+2 raw noise pairs"
            z = self._combine_raw(raw_noises[0], node_mask, B)
            gg = None if self.schedule_gammas is None else torch.as_tensor(np.asarray(self.schedule_gammas), dtype=torch.float32)
            for i, s_ in enumerate(reversed(range(0, T))):
                s_array = torch.full((B, 1), fill_value=s_, device=dev)
                gm = None if gg is None else (gg[s_].expand(B, 1), gg[s_ + 1].expand(B, 1))
                z = self.sample_p_zs_given_zt(s_array / T, (s_array + 1) / T, z, node_mask, edge_mask, context, fix_noise=fix_noise,
                                              mol_shape=N, raw_noise=raw_noises[1 + i], gammas=gm)
            final_raw = tuple(r.to(dev, torch.float32).contiguous() for r in raw_noises[T + 1])
        elif raw_noises is not None:
            assert len(raw_noises) == T + 2, "need T+2 raw noise pairs"
            rx = [r[0].to(dev, torch.float32).contiguous() for r in raw_noises]
            rh = [r[1].to(dev, torch.float32).contiguous() for r in raw_noises]
            _lib.check(lib.hd_noise(h, topo.ptr, rx[0].data_ptr(), rh[0].data_ptr(), rx[0].shape[0], 0, 0, 0, 0,
                                    z.data_ptr(), stream), "hd_noise")
            step_x = torch.stack(rx[1:T + 1]).contiguous()
            step_h = torch.stack(rh[1:T + 1]).contiguous()
            z = run_loop(z, step_x, step_h, step_x.shape[1], 0, 0)
            final_raw = (rx[T + 1], rh[T + 1])
        elif self.noise_mode == "torch" or gnn:
            z = self.sample_combined_position_feature_noise(nb, N, node_mask)
            if nb == 1 and B > 1:
                z = z.expand(B, -1, -1).contiguous()
            em = edge_mask
            for s in reversed(range(0, T)):
                s_array = torch.full((B, 1), fill_value=s, device=dev)
                t_array = s_array + 1
                z = self.sample_p_zs_given_zt(s_array / T, t_array / T, z, node_mask, em, context,
                                              fix_noise=fix_noise, mol_shape=N)
            final_raw = None
        else:
            if z_init is not None:
                z.copy_(z_init)
            else:
                _lib.check(lib.hd_noise(h, topo.ptr, None, None, nb, self.seed, sample_id_base, 0, int(fix_noise),
                                        z.data_ptr(), stream), "hd_noise")
            z = run_loop(z, None, None, nb, self.seed, sample_id_base)
            # decode noise: draw T+1 of the same counter stream, materialised through hd_noise's raw form
            final_raw = "philox"
        self._check_mean_zero(z[:, :, :self.n_dims], node_mask)
        zeros = torch.zeros((B, 1), device=dev)
        eps = self.phi(z, zeros, node_mask, edge_mask, context) if gnn else self.dynamics.forward_with_topology(topo, zeros, z, ctx, None)
        coef3 = tabs["decode"].numpy()
        if final_raw == "philox":
            x, hfeat = self._final_decode(z, eps, node_mask, edge_mask, coef3, fix_noise, None,
                                          philox=(sample_id_base, T + 1))
        else:
            x, hfeat = self._final_decode(z, eps, node_mask, edge_mask, coef3, fix_noise, final_raw)
        return x, hfeat

    @torch.no_grad()
    def sample(self, num_samples, device, context=None, pocket_cond=None, sample_id_base: int = 0):
        """diffusion_qm9.py:347-395: list of {'x': [n_i,3], 'h': [n_i,8], ('context': [n_i,1])} on the CPU."""
        device = torch.device(device)
        sample_n = self.nodes_dist.sample(num_samples)
        pocket = None
        if pocket_cond is not None:
            if not self.pocket:
                raise ValueError("pocket_cond given but the model was built with cfg.pocket = False")
            pocket = (pocket_cond[1].to(device, torch.float32),
                      self.pocket_embed(pocket_cond[0].to(device).long()).to(torch.float32),
                      pocket_cond[2].to(device).bool(), pocket_cond[3].to(device).bool())
        ctx = None
        if context is not None:
            # `zeros([num_samples, n_max, 1]) + context` (:352): a scalar, or any tensor that broadcasts against that shape
            # (e.g. one value per sample as [num_samples, 1, 1]); anything else raises here as it does there
            ctx = torch.zeros([num_samples, max(sample_n), 1]) + torch.as_tensor(context, dtype=torch.float32).cpu()
            if ctx.shape != (num_samples, max(sample_n), 1):
                raise ValueError(f"context of shape {tuple(torch.as_tensor(context).shape)} does not broadcast to "
                                 f"[{num_samples}, {max(sample_n)}, 1]")
        return self._sample_sizes(sample_n, device, None, sample_id_base, pocket, context_full=ctx)

    def _sample_sizes(self, sample_n, device, contexts, sample_id_base, pocket=None, context_full=None):
        """One device batch for the molecule sizes `sample_n` (global sample ids sample_id_base + i); `contexts`: one scalar
        per molecule (merged batches: the value of the batch a molecule belongs to) or None; `context_full`: the
        [num_samples, n_max, 1] tensor of `sample()` instead.  Masks as diffusion_qm9.py:349-353, result slicing as :388-395."""
        num_samples = len(sample_n)
        n_max = max(sample_n)
        sizes = torch.tensor(sample_n)
        ar = torch.arange(n_max)
        node_mask = (ar[None, :] < sizes[:, None]).unsqueeze(-1)
        context = None
        if context_full is not None:
            context = context_full.to(device)
        elif contexts is not None:
            cols = [torch.as_tensor(c, dtype=torch.float32).reshape(-1).cpu() for c in contexts]
            if any(c.numel() != 1 for c in cols):
                raise ValueError("merged batches take one global context value per batch (context_range entries)")
            context = (torch.zeros([num_samples, n_max, 1]) + torch.stack(cols).reshape(num_samples, 1, 1)).to(device)
        node_mask = node_mask.to(device)
        x, h = self.sample_from_masks(node_mask, None, context, sample_id_base=sample_id_base, pocket=pocket)
        x, h = x.cpu(), h.cpu()
        xs = [x[i, :sample_n[i]].clone() for i in range(num_samples)]
        hs = [h[i, :sample_n[i]].clone() for i in range(num_samples)]
        if context is not None:
            ctx = context.cpu()
            return [{'x': xs[i], 'h': hs[i], 'context': ctx[i, :sample_n[i]].clone()} for i in range(num_samples)]
        return [{'x': xs[i], 'h': hs[i]} for i in range(num_samples)]

    def sample_batches(self, batch_size, num_batches, device, context_range=None, protein_data_all=None,
                       sample_id_base: int = 0):
        """diffusion_qm9.py:397-436, incl. the protein branch (`protein_data_all`: list of dicts with
        'residue_type', 'coord', 'pocket_name', 'ligand_name').

        The reference runs the batches one after the other (its shipped job is 16 batches of 2 molecules,
        conf/sample/default.yaml:1-2).  Molecules are independent, and here a sample's bits depend only on its global id
        (counter RNG keyed by the id, per-molecule tiles, batch-size-independent kernels), so consecutive batches are run
        as one device batch of at most `self.merge_batches` molecules / `self.merge_edges` directed edges: same results, bit for bit, as the loop
        (tests/test_gpu_configs.py::test_merged_sample_batches_equal_the_loop), at the throughput of the larger batch.
        The molecule sizes are drawn batch by batch exactly as the loop draws them.  Not merged: the protein branch,
        `noise_mode == "torch"` (torch.randn draws depend on the batch shape), `merge_batches = 0`, and the two
        configurations whose results depend on a batch's padded width (below).  One difference that
        is not a sample's own: the NaN guard (en_dynamics.py:109-111) zeroes the velocity of the whole DEVICE batch."""
        device = torch.device(device)
        # Not merged either: mode 'gnn_dynamics' (torch.randn draws whatever noise_mode says, messages over padded nodes) and
        # aggregation_method 'mean' (the divisor is the padded N of the call) - both depend on the padded width of the batch a
        # molecule sits in - and context_range entries that are not one scalar per batch.
        width_dependent = (getattr(self.dynamics, "mode", "egnn_dynamics") == "gnn_dynamics"
                           or getattr(self.dynamics, "aggregation_method", "sum") == "mean")
        scalar_ctx = context_range is None or all(torch.as_tensor(c).numel() == 1 for c in context_range)
        if (protein_data_all is None and self.merge_batches and self.noise_mode == "philox" and num_batches > 1
                and not width_dependent and scalar_ctx):
            sizes, ctxs = [], []
            for i in range(num_batches):
                sizes.extend(self.nodes_dist.sample(batch_size))
                if context_range is not None:
                    ctxs.extend([context_range[i % len(context_range)]] * batch_size)
            bs = int(batch_size)
            results, lo = [], 0
            while lo < len(sizes):                              # whole batches, greedily, within both limits
                hi, edges = lo + bs, sum(n * (n - 1) for n in sizes[lo:lo + bs])
                while hi < len(sizes):
                    more = sum(n * (n - 1) for n in sizes[hi:hi + bs])
                    if hi + bs - lo > int(self.merge_batches) or (self.merge_edges and edges + more > int(self.merge_edges)):
                        break
                    hi, edges = hi + bs, edges + more
                results.extend(self._sample_sizes(sizes[lo:hi], device, ctxs[lo:hi] if ctxs else None, sample_id_base + lo))
                lo = hi
            return results, []
        protein_cond_all = None
        if protein_data_all is not None:
            protein_cond_all = pocket_tensors(protein_data_all)
        results, test_names = [], []
        for i in range(num_batches):
            lo, hi = i * batch_size, (i + 1) * batch_size
            base = sample_id_base + lo
            if protein_cond_all is not None:
                n_prot = len(protein_cond_all[0])
                cond = [x[lo % n_prot: (hi - 1) % n_prot + 1] for x in protein_cond_all]
                # the reference indexes the names modulo len(protein_cond_all) == 4 (diffusion_qm9.py:428); kept as is
                names = [protein_data_all[k]['pocket_name'] + '/' + protein_data_all[k]['ligand_name']
                         for k in range(lo % len(protein_cond_all), hi % len(protein_cond_all))]
                results.extend(self.sample(batch_size, device, context=None, pocket_cond=cond, sample_id_base=base))
                test_names.extend(names)
            elif context_range is not None:
                results.extend(self.sample(batch_size, device, context=context_range[i % len(context_range)],
                                           pocket_cond=None, sample_id_base=base))
            else:
                results.extend(self.sample(batch_size, device, context=None, pocket_cond=None, sample_id_base=base))
        return results, test_names


class EnVariationalDiffusion(DiffusionQM9):
    """EDM-style entry point (en_diffusion.py:634-667): masks supplied by the caller, fix_noise honoured."""

    @torch.no_grad()
    def sample(self, n_samples, n_nodes, node_mask, edge_mask, context, fix_noise=False):  # type: ignore[override]
        assert node_mask.shape[0] == n_samples and node_mask.shape[1] == n_nodes
        x, h = self.sample_from_masks(node_mask, edge_mask, context, fix_noise=fix_noise)
        if self.debug_checks:
            self._check_mean_zero(x, node_mask)
        max_cog = torch.sum(x, dim=1, keepdim=True).abs().max()
        if self.debug_checks and max_cog.item() > 5e-2:
            print(f'Warning cog drift with error {max_cog.item():.3f}. Projecting the positions down.')
            nm = node_mask.to(torch.float32)
            x = x - (x.sum(1, keepdim=True) / nm.sum(1, keepdim=True)) * nm
        return x, h
