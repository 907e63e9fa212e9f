"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import harness that lets the *unmodified* reference classes (morl-baselines 1.3.0,
mounted read-only at /root/reference) run in the build container, where
gymnasium / mo_gymnasium / wandb / pymoo / cvxpy / cdd are absent.

It pre-populates ``sys.modules`` with minimal stand-ins for the absent third-party
packages (none of them is touched by ``update()``), then imports the reference's
real modules.  It is used by ``tests/golden/make_golden.py`` to produce the golden
fixtures that pin ``oracle/envelope_oracle.py`` and by nothing else.  /root/reference
does not exist on the GPU box, so nothing under ``tests -m gpu``, ``bench.py`` or
``__graft_entry__.smoke()`` may import this file.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MORL_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "morl_baselines"))


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
        self.dtype = dtype
        self._rng = np.random.default_rng(0)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)


class _Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self._rng = np.random.default_rng(0)

    def sample(self):
        return int(self._rng.integers(self.n))


class _MultiBinary:
    def __init__(self, n):
        self.n = int(n)
        self.shape = (self.n,)


class _Env:
    pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs() -> None:
    """Install the stand-in modules (idempotent)."""
    if "gymnasium" in sys.modules and getattr(sys.modules["gymnasium"], "_morl_stub", False):
        return
    sys.dont_write_bytecode = True  # never drop __pycache__ into the reference mount
    spaces = _mod("gymnasium.spaces", Box=_Box, Discrete=_Discrete, MultiBinary=_MultiBinary)
    core = _mod("gymnasium.core", Env=_Env)
    _mod("gymnasium", Env=_Env, spaces=spaces, core=core, Wrapper=_Env, _morl_stub=True)

    class _Cfg(dict):
        def update(self, *a, **k):  # wandb.config.update(...)
            return dict.update(self, *a, **k)

    noop = lambda *a, **k: None  # noqa: E731
    _mod("wandb", log=noop, init=noop, finish=noop, define_metric=noop, Table=noop, Image=noop,
         config=_Cfg(), run=None)
    wr_vec = _mod("mo_gymnasium.wrappers.vector", MOSyncVectorEnv=type("MOSyncVectorEnv", (), {}))
    wr = _mod("mo_gymnasium.wrappers", MONormalizeReward=lambda env, idx=0, **k: env, vector=wr_vec)
    _mod("mo_gymnasium", wrappers=wr, make=noop)

    def _placeholder(*a, **k):
        raise RuntimeError("pymoo placeholder called: not on the update() path")

    _mod("pymoo")
    _mod("pymoo.util")
    _mod("pymoo.util.ref_dirs", get_reference_directions=_placeholder)
    _mod("pymoo.indicators")
    _mod("pymoo.indicators.hv", HV=_placeholder)
    _mod("pymoo.indicators.igd", IGD=_placeholder)
    _mod("pymoo.decomposition")
    _mod("pymoo.decomposition.tchebicheff", Tchebicheff=_placeholder)
    _mod("cvxpy", SolverError=type("SolverError", (Exception,), {}))
    _mod("cdd")
    _mod("seaborn", set_theme=noop, color_palette=noop)
    try:  # distutils is gone from newer Pythons; the reference imports strtobool from it
        import distutils.util  # noqa: F401
    except Exception:  # pragma: no cover
        du = _mod("distutils.util", strtobool=lambda s: int(str(s).lower() in ("1", "true", "yes", "y", "on")))
        _mod("distutils", util=du)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class FakeEnv:
    """MO-Gymnasium-shaped environment stand-in (spaces only; no dynamics)."""

    def __init__(self, obs_dim=32, n_actions=6, reward_dim=3, env_id="fake-minecart-v0", act_dim=None):
        self.observation_space = _Box(-np.inf, np.inf, (obs_dim,))
        if act_dim is None:
            self.action_space = _Discrete(n_actions)
        else:
            self.action_space = _Box(-1.0, 1.0, (act_dim,))
        self.reward_space = _Box(-np.inf, np.inf, (reward_dim,))
        self.reward_dim = reward_dim
        self.unwrapped = self
        self.spec = types.SimpleNamespace(id=env_id)


def import_reference():
    """Return a namespace with the reference's real classes (stubs installed first)."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    install_stubs()
    from morl_baselines.common import buffer, networks, pareto, prioritized_buffer, utils, weights
    from morl_baselines.multi_policy.envelope import envelope

    return types.SimpleNamespace(
        envelope=envelope, buffer=buffer, prioritized_buffer=prioritized_buffer, networks=networks,
        pareto=pareto, weights=weights, utils=utils,
    )


def fill_buffer_synthetic(buf, n, obs_dim, n_actions, reward_dim, seed=0):
    """BASELINE.md section 3 synthetic transitions, in the per-transition draw order stated there."""
    rng = np.random.default_rng(seed)
    for _ in range(n):
        obs = rng.standard_normal(obs_dim).astype(np.float32)
        action = rng.integers(n_actions)
        reward = rng.standard_normal(reward_dim).astype(np.float32)
        next_obs = rng.standard_normal(obs_dim).astype(np.float32)
        done = rng.random() < 0.05
        buf.add(obs, action, reward, next_obs, done)


def import_reference_ac():
    """The continuous-action actor-critic agents (CAPQL, MOSAC, GPI-PD continuous) of the reference, unmodified."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    install_stubs()
    from morl_baselines.multi_policy.capql import capql
    from morl_baselines.multi_policy.gpi_pd import gpi_pd_continuous_action as gpipd_cont
    from morl_baselines.single_policy.ser import mosac_continuous_action as mosac
    from morl_baselines.single_policy.ser import mosac_discrete_action as sacd

    return types.SimpleNamespace(capql=capql, mosac=mosac, gpipd_cont=gpipd_cont, sacd=sacd)
