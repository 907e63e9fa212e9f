"""Weight-conditioned Q network whose parameters live in ONE flat device buffer.

Mirrors the reference ``QNet`` (``multi_policy/envelope/envelope.py:33-77``): same constructor arguments, same
``state_dict`` keys (``net.0.weight``, ``net.0.bias``, ``net.2.weight`` ...), same ``forward(obs, w)`` contract, same
orthogonal initialisation (``common/networks.py:142-157``).  The ``nn.Parameter``s are views into ``self.flat`` laid
out exactly as ``include/morl_hip.h`` describes (W_l row-major (out, in), then b_l), so the HIP kernels and
``torch.save`` / ``load_state_dict`` / ``optim.Adam`` state all see the same bytes.
"""
from __future__ import annotations

from typing import List, Sequence

import torch as th
import torch.nn as nn

from . import ops
from .native import NativeLib


class QNet(nn.Module):
    """Multi-objective Q-network conditioned on the weight vector; forward runs on the HIP kernels."""

    def __init__(self, obs_shape, action_dim: int, rew_dim: int, net_arch: Sequence[int], device="cuda",
                 lib: NativeLib | None = None, max_batch: int = 256, max_weights: int = 64,
                 engine: int | None = None):
        super().__init__()
        if len(obs_shape) != 1:
            raise NotImplementedError("image observations (NatureCNN, envelope.py:50-52) are outside the HIP hot path")
        self.obs_shape = tuple(obs_shape)
        self.action_dim, self.rew_dim = int(action_dim), int(rew_dim)
        self.net_arch = list(net_arch)
        self.feature_extractor = None
        device = th.device(device)
        self.ctx = ops.QNetContext(self.obs_shape[0], self.rew_dim, self.action_dim, self.net_arch, max_batch,
                                   max_weights, lib=lib, fused=engine)
        dims = [self.obs_shape[0] + self.rew_dim] + self.net_arch + [self.action_dim * self.rew_dim]
        # same module structure as common/networks.py:mlp -> identical state_dict keys
        mods: List[nn.Module] = []
        for i in range(len(dims) - 1):
            mods.append(nn.Linear(dims[i], dims[i + 1]))
            if i < len(dims) - 2:
                mods.append(nn.ReLU())
        self.net = nn.Sequential(*mods)
        for m in self.net:  # layer_init: orthogonal gain 1, bias 0
            if isinstance(m, nn.Linear):
                th.nn.init.orthogonal_(m.weight, gain=1)
                th.nn.init.constant_(m.bias, 0.0)
        self.flat = th.empty(self.ctx.n_params, dtype=th.float32, device=device)
        self._bind(self.flat, copy_from_modules=True)

    # -- flat storage ------------------------------------------------------------------------------------------
    def _linears(self):
        return [m for m in self.net if isinstance(m, nn.Linear)]

    def _bind(self, flat: th.Tensor, copy_from_modules: bool) -> None:
        """Make every parameter a view of ``flat`` (optionally copying the current values in first)."""
        with th.no_grad():
            for lin, (wo, wshape, bo, bn) in zip(self._linears(), self.ctx.layer_slices()):
                wv = flat[wo:wo + wshape[0] * wshape[1]].view(wshape)
                bv = flat[bo:bo + bn]
                if copy_from_modules:
                    wv.copy_(lin.weight.detach().to(flat.device))
                    bv.copy_(lin.bias.detach().to(flat.device))
                rg = lin.weight.requires_grad
                lin.weight = nn.Parameter(wv, requires_grad=rg)
                lin.bias = nn.Parameter(bv, requires_grad=rg)
        self.flat = flat

    def _apply(self, fn, recurse=True):
        # .to() / .cuda() / .float(): move the flat buffer, then re-create the views
        new_flat = fn(self.flat)
        if new_flat is not self.flat:
            self._bind(new_flat.contiguous(), copy_from_modules=False)
        return self

    def ordered_parameters(self):
        """Parameters in flat-buffer order (W0, b0, W1, b1, ...) -- also nn.Module.parameters() order."""
        out = []
        for lin in self._linears():
            out += [lin.weight, lin.bias]
        return out

    # -- forward --------------------------------------------------------------------------------------------------
    @th.no_grad()
    def forward(self, obs: th.Tensor, w: th.Tensor) -> th.Tensor:
        """Q(obs_r, w_r) row by row -> (rows, A, R), like ``QNet.forward`` (envelope.py:60-77)."""
        obs = obs.to(self.flat.device, th.float32)
        w = w.to(self.flat.device, th.float32)
        if obs.dim() == 1:
            obs = obs.unsqueeze(0)
        if w.dim() == 1:
            w = w.unsqueeze(0).expand(obs.size(0), -1)
        obs, w = obs.contiguous(), w.contiguous()
        n = obs.size(0)
        if n > self.ctx.max_batch:
            self._grow(n, self.ctx.max_weights)
        return ops.qnet_forward_rows(self.ctx, self.flat, obs, w)

    def _grow(self, max_batch: int, max_weights: int) -> None:
        old = self.ctx
        self.ctx = ops.QNetContext(old.obs_dim, old.reward_dim, old.n_actions, old.net_arch,
                                   max(max_batch, old.max_batch), max(max_weights, old.max_weights), lib=old.lib,
                                   fused=old.engine)
        old.close()

    def ensure_capacity(self, batch: int, weights: int) -> None:
        if batch > self.ctx.max_batch or weights > self.ctx.max_weights:
            self._grow(batch, weights)
