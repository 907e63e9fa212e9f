"""Host-side module shells of the actor-critic agents.

The ``nn.Module``s here exist for the drop-in surface only -- ``state_dict()`` keys, ``parameters()`` order and the
torch-RNG consumption of construction are those of the reference (``common/networks.py:10-48`` ``mlp``, ``:142-157``
``layer_init``) -- while every parameter is a VIEW into the flat device buffers of ``ACEngine``; no forward pass ever
runs through these modules (the HIP engine reads the flat buffers directly).
"""
from __future__ import annotations

from typing import List, Sequence

import os

import numpy as np
import torch as th
from torch import nn


def build_mlp(input_dim: int, output_dim: int, net_arch: Sequence[int], drop_rate: float = 0.0,
              layer_norm: bool = False) -> nn.Sequential:
    """Same module sequence as the reference's ``mlp`` (so ``net.<k>.weight`` keys line up)."""
    assert len(net_arch) > 0
    mods: List[nn.Module] = []
    d = input_dim
    for h in net_arch:
        mods.append(nn.Linear(d, h))
        if drop_rate > 0.0:
            mods.append(nn.Dropout(p=drop_rate))
        if layer_norm:
            mods.append(nn.LayerNorm(h))
        mods.append(nn.ReLU())
        d = h
    if output_dim > 0:
        mods.append(nn.Linear(d, output_dim))
    return nn.Sequential(*mods)


@th.no_grad()
def layer_init(layer, weight_gain: float = 1, bias_const: float = 0) -> None:
    if isinstance(layer, nn.Linear):
        th.nn.init.orthogonal_(layer.weight, gain=weight_gain)
        th.nn.init.constant_(layer.bias, bias_const)


class QNetworkShell(nn.Module):
    """``QNetwork`` of capql.py:161-173 / gpi_pd_continuous_action.py:61-73: ``self.net``."""

    def __init__(self, in_dim, rew_dim, net_arch, drop_rate=0.0, layer_norm=False):
        super().__init__()
        self.net = build_mlp(in_dim, rew_dim, net_arch, drop_rate=drop_rate, layer_norm=layer_norm)
        self.apply(layer_init)


class SoftQShell(nn.Module):
    """``MOSoftQNetwork`` of mosac_continuous_action.py:28-57: ``self.critic``."""

    def __init__(self, in_dim, rew_dim, net_arch):
        super().__init__()
        self.critic = build_mlp(in_dim, rew_dim, net_arch)
        self.apply(layer_init)


class PolicyShell(nn.Module):
    """``latent_pi`` trunk + named heads, plus the action rescaling buffers."""

    def __init__(self, in_dim, act_dim, net_arch, head_names: Sequence[str], action_low, action_high):
        super().__init__()
        self.latent_pi = build_mlp(in_dim, -1, net_arch)
        for name in head_names:
            setattr(self, name, nn.Linear(net_arch[-1], act_dim))
        self.register_buffer("action_scale", th.tensor((action_high - action_low) / 2.0, dtype=th.float32))
        self.register_buffer("action_bias", th.tensor((action_high + action_low) / 2.0, dtype=th.float32))
        self.apply(layer_init)


def bind(module: nn.Module, views: Sequence[th.Tensor], copy_in: bool = True) -> None:
    """Make the module's parameters views of the engine's flat buffers (``views`` in ``parameters()`` order)."""
    params = list(module.parameters())
    assert len(params) == len(views), (len(params), len(views))
    with th.no_grad():
        for p, v in zip(params, views):
            assert tuple(p.shape) == tuple(v.shape), (p.shape, v.shape)
            if copy_in:
                v.copy_(p.detach().to(v.device))
            p.data = v
            p.requires_grad_(False)


def adam_state_dict(params_views, m_views, v_views, step: int, lr: float) -> dict:
    """torch.optim.Adam.state_dict() layout built from the flat moment buffers (so reference code can load it)."""
    state = {}
    if step > 0:
        for i, (m, v) in enumerate(zip(m_views, v_views)):
            state[i] = {"step": th.tensor(float(step)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
    group = {"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(params_views)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(sd: dict, m_views, v_views) -> int:
    """Inverse of ``adam_state_dict``; returns the step count (0 if the optimiser had not stepped)."""
    step = 0
    with th.no_grad():
        for i, (m, v) in enumerate(zip(m_views, v_views)):
            st = sd["state"].get(i)
            if st is None:
                m.zero_()
                v.zero_()
                continue
            m.copy_(st["exp_avg"].to(m.device))
            v.copy_(st["exp_avg_sq"].to(v.device))
            step = int(float(st["step"]))
    return step


# Where the exploration / target / model noise is drawn.  Default: on the device that consumes it (its own generator).
# HOST_NOISE = True (or MORL_HOST_NOISE=1) draws every sample from torch's CPU generator -- the stream the reference consumes
# when it runs on CPU -- and copies it over: a seeded GPU run then replays the reference's seeded CPU run draw for draw
# (tests/test_train_traces.py on the MI355X).  A few KB per update; the training loops are host-bound anyway.
HOST_NOISE = os.environ.get("MORL_HOST_NOISE", "0") == "1"


def noise_device(device) -> th.device:
    return th.device("cpu") if HOST_NOISE else th.device(device)


def randn(shape, device) -> th.Tensor:
    """Standard-normal draws of ``shape`` for ``device`` (see HOST_NOISE)."""
    return th.randn(tuple(shape), dtype=th.float32, device=noise_device(device)).to(device)


def as_f32(x, device) -> th.Tensor:
    if isinstance(x, np.ndarray):
        x = th.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return x.to(device=device, dtype=th.float32)
