"""Seeded inputs for the golden fixtures (numpy RNG only, so they regenerate identically everywhere).

Shared by ``make_golden.py`` (which feeds them to the *reference* in the build container) and by the
parity tests (which feed them to the oracle and to the HIP path).  No reference import here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass(frozen=True)
class Case:
    name: str
    B: int
    W: int
    D: int
    A: int
    R: int
    arch: tuple
    envelope: bool = True
    homotopy_lambda: float = 0.0
    max_grad_norm: Optional[float] = 1.0
    step: int = 1            # 1-based Adam step taken by this update
    gamma: float = 0.99
    lr: float = 3e-4
    dup_weights: bool = False
    seed: int = 0
    subsample: int = 1        # stride used when storing large per-parameter outputs


CASES: List[Case] = [
    Case("tiny_mse", B=8, W=4, D=5, A=3, R=2, arch=(16, 16)),
    Case("small_homotopy", B=16, W=4, D=7, A=6, R=3, arch=(32, 32, 32), homotopy_lambda=0.3,
         max_grad_norm=0.1, step=7, gamma=0.98, seed=1),
    Case("ddqn", B=8, W=3, D=6, A=4, R=3, arch=(24, 24), envelope=False, seed=2),
    Case("noclip_r4", B=8, W=2, D=4, A=2, R=4, arch=(16,), max_grad_norm=None, step=3, seed=3),
    Case("dup_weights", B=8, W=5, D=6, A=3, R=3, arch=(16, 16), dup_weights=True, seed=4),
    Case("flagship_b32w8", B=32, W=8, D=32, A=6, R=3, arch=(256, 256, 256, 256), step=2, seed=5, subsample=101),
]

# The BASELINE.json shape itself (B=256 x W=64 x R=3).  Kept out of CASES: the emulator cannot run it and its target
# tensor is large; tests/test_flagship_golden.py (gpu) uses the reduced fixture written by make_golden.py.
FLAGSHIP = Case("flagship_full", B=256, W=64, D=32, A=6, R=3, arch=(256, 256, 256, 256), step=3, seed=6, subsample=257)
# BASELINE.json configs[1]: Envelope on mo-minecart-v0 (7 observations, 6 actions, 3 objectives), batch 256 x 32 weights
MINECART = Case("minecart_b256w32", B=256, W=32, D=7, A=6, R=3, arch=(256, 256, 256, 256), step=2, seed=8, gamma=0.98,
                max_grad_norm=0.1, subsample=257)
FULL_SIZE = [FLAGSHIP, MINECART]


def layer_dims(c: Case):
    return [c.D + c.R] + list(c.arch) + [c.A * c.R]


def make_inputs(c: Case) -> dict:
    """Everything one ``Envelope.update()`` iteration consumes, as numpy arrays."""
    rng = np.random.default_rng(1000 + c.seed)
    dims = layer_dims(c)
    online, target = [], []
    for i in range(len(dims) - 1):
        w = (rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)
        b = (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)
        online += [w, b]
        target += [(w + 0.05 * rng.standard_normal(w.shape).astype(np.float32) / np.sqrt(dims[i])).astype(np.float32),
                   (b + 0.02 * rng.standard_normal(b.shape)).astype(np.float32)]
    if c.step > 1:
        exp_avg = [(1e-3 * rng.standard_normal(p.shape)).astype(np.float32) for p in online]
        exp_avg_sq = [(1e-6 * rng.random(p.shape)).astype(np.float32) for p in online]
    else:
        exp_avg = [np.zeros_like(p) for p in online]
        exp_avg_sq = [np.zeros_like(p) for p in online]
    obs = rng.standard_normal((c.B, c.D)).astype(np.float32)
    actions = rng.integers(c.A, size=(c.B, 1)).astype(np.uint8)
    rewards = rng.standard_normal((c.B, c.R)).astype(np.float32)
    next_obs = rng.standard_normal((c.B, c.D)).astype(np.float32)
    dones = (rng.random((c.B, 1)) < 0.25).astype(np.float32)
    sw = rng.standard_normal((c.W, c.R))
    sw = np.abs(sw) / np.linalg.norm(sw, ord=1, axis=1, keepdims=True)  # what random_weights("gaussian") yields
    if c.dup_weights:
        sw[3] = sw[1]
        sw[4] = sw[0]
    return dict(online=online, target=target, exp_avg=exp_avg, exp_avg_sq=exp_avg_sq, obs=obs, actions=actions,
                rewards=rewards, next_obs=next_obs, dones=dones, sampled_w=sw)


# ---- Pareto adversarial sets (float64) -------------------------------------------------
def pareto_sets() -> dict:
    rng = np.random.default_rng(77)
    sets = {}
    a = rng.random((40, 3))
    a[7] = a[3]
    a[21] = a[3]                     # exact duplicates of a (probably) non-dominated point
    a[30] = a[12] * 0.5              # strictly dominated
    a[31] = a[30]                    # duplicated dominated point
    sets["dups"] = a
    t = np.round(rng.random((60, 2)) * 4) / 4    # many ties per coordinate
    sets["ties2d"] = t
    z = rng.standard_normal((30, 3))
    z[0] = [0.0, 1.0, -1.0]
    z[1] = [-0.0, 1.0, -1.0]         # -0.0 == 0.0 -> duplicates
    sets["signed_zero"] = z
    sets["pair_dom"] = np.array([[1.0, 2.0], [2.0, 3.0]])
    sets["pair_equal"] = np.array([[1.0, 2.0], [1.0, 2.0]])
    sets["pair_incomparable"] = np.array([[1.0, 3.0], [2.0, 2.0]])
    sets["r4_200"] = rng.random((200, 4)) ** 2
    sets["all_same"] = np.ones((9, 3))
    chain = np.arange(12, dtype=np.float64)[:, None] * np.ones((1, 3))
    sets["chain"] = chain[::-1].copy()
    return sets
