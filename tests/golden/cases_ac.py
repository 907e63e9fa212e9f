"""Seeded inputs of the actor-critic golden cases (CAPQL, MOSAC, GPI-PD continuous).

Shared by ``make_golden_ac.py`` (which runs the unmodified reference on them), the oracle tests and the kernel parity
tests, so all three see the same parameters, batches and random draws.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple

import numpy as np
import torch as th

import ac_oracle as ac


@dataclass(frozen=True)
class ACCase:
    name: str
    algo: str                      # "capql" | "mosac" | "gpipd"
    D: int
    Ad: int
    R: int
    arch: Tuple[int, ...]
    B: int
    step: int = 1                  # 1-based Adam step taken by this update (> 1: optimiser state is pre-seeded)
    gamma: float = 0.99
    tau: float = 0.005
    lr: float = 3e-4
    alpha: float = 0.2
    low: float = -1.0
    high: float = 1.0
    seed: int = 0
    subsample: int = 1
    # mosac
    autotune: bool = True
    policy_freq: int = 2
    global_step: int = 100         # % policy_freq == 0 -> actor update; always a multiple of target_net_freq (1)
    q_lr: float = 1e-3
    log_alpha0: float = 0.0
    # gpipd
    n_support: int = 1
    per: bool = False
    n_updates: int = 0             # % delay_policy_update (2) == 0 -> policy update
    drop_rate: float = 0.01
    layer_norm: bool = True
    extra: dict = field(default_factory=dict)


AC_CASES = [
    ACCase("capql_small", "capql", D=11, Ad=3, R=2, arch=(64, 64), B=32),
    ACCase("capql_step5", "capql", D=7, Ad=2, R=3, arch=(48, 32), B=20, step=5, low=-2.0, high=1.0, alpha=0.05, seed=3),
    ACCase("capql_cheetah", "capql", D=17, Ad=6, R=2, arch=(256, 256), B=128, seed=5, subsample=7),
    ACCase("mosac_small", "mosac", D=11, Ad=3, R=2, arch=(64, 64), B=32, seed=11),
    ACCase("mosac_noauto_odd", "mosac", D=6, Ad=2, R=3, arch=(32, 48), B=24, step=4, autotune=False, global_step=101,
           low=-0.5, high=2.0, seed=12),
    ACCase("mosac_hopper", "mosac", D=11, Ad=3, R=3, arch=(256, 256), B=128, step=3, seed=13, subsample=7,
           log_alpha0=-0.7),
    ACCase("sacd_small", "sacd", D=9, Ad=4, R=2, arch=(32, 32), B=24, seed=31, tau=1.0, q_lr=3e-4),
    ACCase("sacd_step3_noauto", "sacd", D=6, Ad=3, R=3, arch=(24, 32), B=16, step=3, autotune=False, seed=32, tau=0.3,
           q_lr=1e-3, log_alpha0=0.0),
    ACCase("sacd_minecart", "sacd", D=7, Ad=6, R=3, arch=(256, 256), B=128, step=2, seed=33, tau=1.0, q_lr=3e-4,
           subsample=7, log_alpha0=-0.4),
    ACCase("gpipd_small", "gpipd", D=11, Ad=3, R=2, arch=(64, 64), B=32, seed=21),
    ACCase("gpipd_support_per", "gpipd", D=7, Ad=2, R=3, arch=(48, 32), B=16, step=3, n_support=3, per=True, seed=22,
           low=-2.0, high=1.0),
    ACCase("gpipd_nopolicy_plain", "gpipd", D=5, Ad=2, R=2, arch=(32, 32), B=16, n_updates=1, drop_rate=0.0,
           layer_norm=False, seed=23),
    ACCase("gpipd_hopper", "gpipd", D=11, Ad=3, R=3, arch=(256, 256), B=128, n_support=4, per=True, seed=24,
           subsample=7),
]


def specs(c: ACCase):
    """(q-net spec, policy-trunk spec) of the case."""
    if c.algo == "capql":
        return (ac.MlpSpec(c.D + c.Ad + c.R, c.arch, c.R), ac.MlpSpec(c.D + c.R, c.arch))
    if c.algo == "mosac":
        return (ac.MlpSpec(c.D + c.Ad, c.arch, c.R), ac.MlpSpec(c.D, c.arch))
    if c.algo == "sacd":                      # Ad = number of discrete actions; the "trunk" + one head of A logits
        return (ac.MlpSpec(c.D, c.arch, c.Ad * c.R), ac.MlpSpec(c.D, c.arch))
    return (ac.MlpSpec(c.D + c.Ad + c.R, c.arch, c.R, layer_norm=c.layer_norm, drop_rate=c.drop_rate),
            ac.MlpSpec(c.D + c.R, c.arch))


def n_heads(c: ACCase) -> int:
    return 1 if c.algo in ("gpipd", "sacd") else 2


def _perturbed(ps, rng, scale=0.05):
    """Biases / LayerNorm affine away from their (0, 1) initial values so those paths carry signal."""
    out = []
    for p in ps:
        if p.dim() == 1:
            p = p + th.tensor(rng.standard_normal(p.shape).astype(np.float32) * scale)
        out.append(p.contiguous())
    return out


def make_inputs(c: ACCase) -> dict:
    rng = np.random.default_rng(1000 + c.seed)
    qspec, trunk = specs(c)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731

    def dense(shape):
        # numpy-only draws (bit-reproducible on every machine; torch's orthogonal_ goes through LAPACK, whose result
        # depends on the host CPU / thread count, and the fixtures must be regenerable anywhere)
        return th.tensor(f32(rng.standard_normal(shape) / np.sqrt(shape[1])))

    def net(spec):
        return _perturbed([dense(s) if len(s) == 2 else (th.ones(s) if k % 4 == 2 and spec.layer_norm else th.zeros(s))
                           for k, s in enumerate(spec.shapes())], rng)

    head = lambda: [dense((c.Ad, c.arch[-1])), th.tensor(f32(rng.standard_normal(c.Ad) * 0.05))]  # noqa: E731
    q = [net(qspec) for _ in range(2)]
    tq = [[p + th.tensor(f32(rng.standard_normal(p.shape) * 0.01)) for p in net_] for net_ in q]
    pol = net(trunk)
    for _ in range(n_heads(c)):
        pol += head()
    inp = dict(q=q, tq=tq, pol=pol)
    if c.algo == "gpipd":
        inp["tpol"] = [p + th.tensor(f32(rng.standard_normal(p.shape) * 0.01)) for p in pol]

    def opt_state(ps):
        if c.step <= 1:
            return dict(exp_avg=[th.zeros_like(p) for p in ps], exp_avg_sq=[th.zeros_like(p) for p in ps])
        return dict(exp_avg=[th.tensor(f32(rng.standard_normal(p.shape) * 1e-3)) for p in ps],
                    exp_avg_sq=[th.tensor(f32(rng.random(p.shape) * 1e-5)) for p in ps])

    inp["q_state"] = opt_state([p for net in q for p in net])
    inp["p_state"] = opt_state(pol)
    nrows = c.B * (2 if (c.algo == "gpipd" and c.n_support > 1) else 1)
    inp["obs"] = f32(rng.standard_normal((c.B, c.D)))
    inp["actions"] = (f32(rng.integers(0, c.Ad, (c.B, 1))) if c.algo == "sacd" else
                      f32(rng.uniform(c.low, c.high, (c.B, c.Ad))))
    inp["rewards"] = f32(rng.standard_normal((c.B, c.R)))
    inp["next_obs"] = f32(rng.standard_normal((c.B, c.D)))
    inp["dones"] = f32(rng.random((c.B, 1)) < 0.1)
    wv = np.abs(rng.standard_normal((c.B, c.R)))
    inp["w"] = f32(wv / wv.sum(1, keepdims=True))             # CAPQL: one weight per stored transition
    inp["eps_next"] = f32(rng.standard_normal((nrows, c.Ad)))
    inp["eps_pi"] = [f32(rng.standard_normal((c.B, c.Ad))) for _ in range(max(1, c.policy_freq))]
    inp["eps_alpha"] = [f32(rng.standard_normal((c.B, c.Ad))) for _ in range(max(1, c.policy_freq))]
    if c.algo in ("mosac", "sacd"):
        wv = np.abs(rng.standard_normal(c.R))
        inp["weights"] = f32(wv / wv.sum())
        inp["al_state"] = (dict(exp_avg=[th.zeros(1)], exp_avg_sq=[th.zeros(1)]) if c.step <= 1 else
                           dict(exp_avg=[th.tensor([2e-3])], exp_avg_sq=[th.tensor([3e-6])]))
    if c.algo == "gpipd":
        sup = np.abs(rng.standard_normal((max(1, c.n_support), c.R)))
        inp["support"] = f32(sup / sup.sum(1, keepdims=True))
        wv = np.abs(rng.standard_normal(c.R))
        inp["weight"] = f32(wv / wv.sum())
        inp["choice"] = rng.integers(0, max(1, c.n_support), c.B)   # stands in for random.choices (pinned)
        keep = lambda h: f32(rng.random((nrows, h)) >= c.drop_rate)  # noqa: E731
        inp["drop"] = ({k: [[keep(h) for h in c.arch] for _ in range(2)] for k in ("target", "q", "q_pi")}
                       if c.drop_rate > 0 else {})
    inp["scale"] = th.full((c.Ad,), (c.high - c.low) / 2.0)
    inp["bias"] = th.full((c.Ad,), (c.high + c.low) / 2.0)
    return inp
