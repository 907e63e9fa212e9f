"""A self-contained multi-objective MDP with the MO-Gymnasium step/reset signature, plus an exact hypervolume.

TEST INFRASTRUCTURE.  The container has no gymnasium / mo_gymnasium / pymoo, so the north star's "Pareto-front
hypervolume within 1 % of the reference after equal gradient steps" is checked on this environment: the unmodified
reference agent is trained on it in the build container (``tests/golden/make_golden.py`` -> ``train_trace.npz``) and
the HIP agent is trained on it on the GPU with the same seeds (``tests/test_train_hv.py``).

``TreasureLine`` is a small deep-sea-treasure-like grid: a submarine starts at the surface (row 0, column 0); column c
has its sea floor -- and a treasure -- at row c + 1.  Objectives: (treasure value, -1 per step).  Treasure values are
concave in the number of steps needed, so every treasure is on the convex hull of the Pareto front and linear
scalarisation (what Envelope / GPI learn) can reach each of them.
"""
from __future__ import annotations

import sys
import types

import numpy as np


def _space_base(name):
    """The reference tells spaces apart with isinstance(gymnasium.spaces.X); subclass X when a gymnasium
    (real or the golden generator's stand-in) is already imported, else plain object."""
    gs = sys.modules.get("gymnasium.spaces")
    return getattr(gs, name, object) if gs is not None else object


class DiscreteSpace(_space_base("Discrete")):
    def __init__(self, n, seed=0):
        self.n = int(n)
        self.shape = ()
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(self.n))


class BoxSpace(_space_base("Box")):
    def __init__(self, low, high, shape, seed=0):
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=np.float32)
        self.high = np.full(self.shape, high, dtype=np.float32)
        self.dtype = np.float32
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(np.float32)


class TreasureLine:
    """obs = one-hot(row) ++ one-hot(col); actions 0 up, 1 down, 2 left, 3 right; 2 objectives."""

    COLS = 4
    ROWS = 5
    VALUES = (1.0, 2.6, 3.8, 4.6)      # reached after 1, 3, 5, 7 steps
    HORIZON = 16

    def __init__(self, seed=0, env_id="treasure-line-v0"):
        self.observation_space = BoxSpace(0.0, 1.0, (self.ROWS + self.COLS,), seed)
        self.action_space = DiscreteSpace(4, seed)
        self.reward_space = BoxSpace(-1.0, 5.0, (2,), seed)
        self.reward_dim = 2
        self.unwrapped = self
        self.spec = types.SimpleNamespace(id=env_id)
        self.r = self.c = self.t = 0
        self.action_log = []           # every action ever taken (stream-parity checks)

    def _obs(self):
        o = np.zeros(self.ROWS + self.COLS, dtype=np.float32)
        o[self.r] = 1.0
        o[self.ROWS + self.c] = 1.0
        return o

    def reset(self, seed=None, options=None):
        self.r = self.c = self.t = 0
        return self._obs(), {}

    def close(self):
        pass

    def step(self, action):
        action = int(action)
        self.action_log.append(action)
        r, c = self.r, self.c
        if action == 0:
            r -= 1
        elif action == 1:
            r += 1
        elif action == 2:
            c -= 1
        else:
            c += 1
        # stay inside the water column: rows 0 .. floor(c), columns 0 .. COLS-1
        if 0 <= c < self.COLS and 0 <= r <= c + 1:
            self.r, self.c = r, c
        self.t += 1
        terminated = self.r == self.c + 1
        reward = np.array([self.VALUES[self.c] if terminated else 0.0, -1.0], dtype=np.float32)
        truncated = (not terminated) and self.t >= self.HORIZON
        return self._obs(), reward, terminated, truncated, {}


class PointReach:
    """Continuous-action counterpart: a point on a line segment; action in [-1, 1] moves it.

    obs = (x, t / H); objectives: (closeness to +1 end, closeness to -1 end) paid every step, so the preferred end --
    and the speed of going there -- depends on the weight vector.  Used by the CAPQL / MOSAC training checks.
    """

    HORIZON = 12

    def __init__(self, seed=0, env_id="point-reach-v0"):
        self.observation_space = BoxSpace(-1.0, 1.0, (2,), seed)
        self.action_space = BoxSpace(-1.0, 1.0, (1,), seed)
        self.reward_space = BoxSpace(0.0, 1.0, (2,), seed)
        self.reward_dim = 2
        self.unwrapped = self
        self.spec = types.SimpleNamespace(id=env_id)
        self.x = 0.0
        self.t = 0
        self.action_log = []

    def _obs(self):
        return np.array([self.x, self.t / self.HORIZON], dtype=np.float32)

    def reset(self, seed=None, options=None):
        self.x, self.t = 0.0, 0
        return self._obs(), {}

    def close(self):
        pass

    def step(self, action):
        self.action_log.append(np.asarray(action, dtype=np.float32).reshape(-1).copy())
        a = float(np.clip(np.asarray(action, dtype=np.float64).reshape(-1)[0], -1.0, 1.0))
        self.x = float(np.clip(self.x + 0.25 * a, -1.0, 1.0))
        self.t += 1
        reward = np.array([0.5 * (1.0 + self.x), 0.5 * (1.0 - self.x) * 0.8], dtype=np.float32)
        truncated = self.t >= self.HORIZON
        return self._obs(), reward, False, truncated, {}


def hypervolume_2d(points, ref) -> float:
    """Exact hypervolume (maximisation) of 2-objective points w.r.t. ``ref``.

    Same quantity as the reference's ``performance_indicators.hypervolume`` (``performance_indicators.py:15-25``:
    pymoo ``HV(ref_point * -1)(points * -1)``) -- the area dominated by the points and dominating ``ref``.
    """
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    ref = np.asarray(ref, dtype=np.float64)
    pts = pts[(pts > ref).all(axis=1)]
    if len(pts) == 0:
        return 0.0
    order = np.argsort(-pts[:, 0], kind="stable")
    hv, best_y = 0.0, ref[1]
    for x, y in pts[order]:
        if y > best_y:
            hv += (x - ref[0]) * (y - best_y)
            best_y = y
    return float(hv)


def equally_spaced_weights_2d(n: int) -> np.ndarray:
    """The 2-objective case of ``common/weights.py:38-48`` (uniform simplex lattice)."""
    a = np.linspace(0.0, 1.0, n)
    return np.stack([a, 1.0 - a], axis=1)


def greedy_front(agent, env, weights, gamma_attr="gamma"):
    """Discounted vector return of one greedy episode per weight (``evaluation.py:23-67`` semantics)."""
    front = []
    for w in weights:
        obs, _ = env.reset()
        done, g = False, 1.0
        ret = np.zeros(len(w), dtype=np.float64)
        while not done:
            obs, r, term, trunc, _ = env.step(agent.eval(obs, np.asarray(w, dtype=np.float32)))
            done = term or trunc
            ret = ret + g * np.asarray(r, dtype=np.float64)
            g *= getattr(agent, gamma_attr)
        front.append(ret)
    return np.asarray(front)
