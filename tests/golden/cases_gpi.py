"""Seeded inputs of the GPI-PD (discrete actions) golden cases -- shared by the generator, the oracle tests and the
kernel parity tests.  numpy-only draws (bit-reproducible on every machine)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch as th

import gpi_oracle as go


@dataclass(frozen=True)
class GpiCase:
    name: str
    D: int
    A: int
    R: int
    arch: Tuple[int, ...]
    B: int
    n_support: int = 1
    gpi_pd: bool = True
    per: bool = True
    step: int = 1
    layer_norm: bool = True
    drop_rate: float = 0.01
    max_grad_norm: float = -1.0          # < 0: None
    gamma: float = 0.99
    lr: float = 3e-4
    min_priority: float = 0.01
    seed: int = 0
    subsample: int = 1


GPI_CASES = [
    GpiCase("gpi_small", D=6, A=3, R=2, arch=(32, 32, 32), B=16, n_support=3),
    GpiCase("gpi_single_support", D=5, A=4, R=3, arch=(24, 32), B=12, n_support=1, gpi_pd=True, step=4, seed=1),
    GpiCase("gpi_many_support_clip", D=7, A=3, R=2, arch=(32, 24, 32), B=10, n_support=7, max_grad_norm=0.5, seed=2),
    GpiCase("gpi_ls_plain", D=4, A=2, R=2, arch=(16, 16), B=8, n_support=2, gpi_pd=False, layer_norm=False,
            drop_rate=0.0, seed=3),
    GpiCase("gpi_minecart", D=7, A=6, R=3, arch=(256, 256, 256, 256), B=128, n_support=4, seed=4, subsample=11),
]


def spec_of(c: GpiCase) -> go.GpiSpec:
    return go.GpiSpec(c.D, c.R, c.A, c.arch, c.layer_norm, c.drop_rate)


def make_inputs(c: GpiCase) -> dict:
    rng = np.random.default_rng(5000 + c.seed)
    spec = spec_of(c)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731

    def net():
        ps = []
        for k, s in enumerate(spec.shapes()):
            if len(s) == 2:
                ps.append(th.tensor(f32(rng.standard_normal(s) / np.sqrt(s[1]))))
            else:
                is_gamma = c.layer_norm and k >= 4 and (k - 4) % 4 == 2 and k < len(spec.shapes()) - 2
                base = 1.0 if is_gamma else 0.0
                ps.append(th.tensor(f32(base + rng.standard_normal(s) * 0.05)))
        return ps

    q = [net() for _ in range(2)]
    tq = [[p + th.tensor(f32(rng.standard_normal(p.shape) * 0.01)) for p in n] for n in q]
    flat = [p for n in q for p in n]
    if c.step <= 1:
        state = dict(exp_avg=[th.zeros_like(p) for p in flat], exp_avg_sq=[th.zeros_like(p) for p in flat])
    else:
        state = dict(exp_avg=[th.tensor(f32(rng.standard_normal(p.shape) * 1e-3)) for p in flat],
                     exp_avg_sq=[th.tensor(f32(rng.random(p.shape) * 1e-5)) for p in flat])
    rows = c.B * (2 if c.n_support > 1 else 1)
    inp = dict(q=q, tq=tq, state=state)
    inp["obs"] = f32(rng.standard_normal((c.B, c.D)))
    inp["actions"] = rng.integers(0, c.A, (c.B, 1)).astype(np.uint8)
    inp["rewards"] = f32(rng.standard_normal((c.B, c.R)))
    inp["next_obs"] = f32(rng.standard_normal((c.B, c.D)))
    inp["dones"] = f32(rng.random((c.B, 1)) < 0.1)
    sup = np.abs(rng.standard_normal((c.n_support, c.R)))
    inp["support"] = f32(sup / sup.sum(1, keepdims=True))
    wv = np.abs(rng.standard_normal(c.R))
    inp["weight"] = f32(wv / wv.sum())
    inp["choice"] = rng.integers(0, c.n_support, c.B)            # random.choices stand-in
    inp["sample4"] = rng.permutation(c.n_support)[:4]            # random.sample stand-in (|M| > 5)
    K = 5 if c.n_support > 5 else c.n_support
    hidden = c.arch[1:]
    keep = lambda n_rows: [[f32(rng.random((n_rows, h)) >= c.drop_rate) for h in hidden] for _ in range(2)]  # noqa: E731
    inp["drop"] = dict(target=keep(rows), env=keep(rows * K), q=keep(rows)) if c.drop_rate > 0 else {}
    inp["K"] = K
    return inp


def rows_and_weights(c: GpiCase, inp):
    """Doubled batch, per-row weights and the sampled weight set of gpi_pd.py:425-444."""
    T = th.tensor
    batch = [T(inp["obs"]), T(inp["actions"].astype(np.float32)), T(inp["rewards"]), T(inp["next_obs"]), T(inp["dones"])]
    weight = T(inp["weight"])
    sup = [T(s) for s in inp["support"]]
    if c.n_support > 1:
        batch = [b.repeat(2, 1) for b in batch]
        w = th.vstack([weight] * c.B + [sup[i] for i in inp["choice"][:c.B]])
    else:
        w = weight.repeat(c.B, 1)
    if c.n_support > 5:
        sampled_w = th.stack([weight] + [sup[i] for i in inp["sample4"]])
    else:
        sampled_w = th.stack(sup)
    return batch, w, sampled_w
