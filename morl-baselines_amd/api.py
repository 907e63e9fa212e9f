"""Base classes of the drop-in boundary: ``MOPolicy`` / ``MOAgent``.

When the reference package ``morl_baselines`` (and its gymnasium / wandb dependencies) is importable, its
*own* base classes are used unchanged (``common/morl_algorithm.py:23-221`` and ``:224-337``) -- that is the
drop-in boundary.  On machines where it is not installed (the GPU test box has no gymnasium / wandb), minimal
mirrors with the same names, signatures and attributes are provided so the HIP agents still construct and
train; evaluation helpers that need the reference's ``common/evaluation.py`` raise a clear ImportError there.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional, Union

import numpy as np
import torch as th

try:  # the real thing, unchanged
    from morl_baselines.common.morl_algorithm import MOAgent, MOPolicy  # type: ignore  # noqa: F401

    HAVE_REFERENCE_API = True
except Exception:  # pragma: no cover - exercised on boxes without the reference installed
    HAVE_REFERENCE_API = False

    class MOPolicy(ABC):
        """Mirror of ``morl_algorithm.py:23-221`` (constructor, abstract eval/update, buffer/net accessors)."""

        def __init__(self, id: Optional[int] = None, device: Union[th.device, str] = "auto") -> None:
            self.id = id
            self.device = th.device("cuda" if th.cuda.is_available() else "cpu") if device == "auto" else device
            self.global_step = 0

        @abstractmethod
        def eval(self, obs: np.ndarray, w: Optional[np.ndarray]) -> Union[int, np.ndarray]:
            ...

        @abstractmethod
        def update(self) -> None:
            ...

        def policy_eval(self, eval_env, num_episodes: int = 5, scalarization=np.dot,
                        weights: Optional[np.ndarray] = None, log: bool = False):
            """Runs ``num_episodes`` greedy episodes (``morl_algorithm.py:85-125`` -> ``evaluation.py:118-144``)."""
            rets = [_eval_mo(self, eval_env, weights, scalarization) for _ in range(num_episodes)]
            return tuple(np.mean([r[k] for r in rets], axis=0) for k in range(4))

        def get_policy_net(self):
            pass

        def get_buffer(self):
            pass

        def set_buffer(self, buffer):
            pass

        def get_save_dict(self, save_replay_buffer: bool = False) -> dict:
            pass

        def load(self, path, load_replay_buffer=True):
            pass

        def set_weights(self, weights: np.ndarray):
            pass

    class MOAgent(ABC):
        """Mirror of ``morl_algorithm.py:224-337`` (env feature extraction, seeding)."""

        def __init__(self, env, device: Union[th.device, str] = "auto", seed: Optional[int] = None) -> None:
            self.extract_env_info(env)
            self.device = th.device("cuda" if th.cuda.is_available() else "cpu") if device == "auto" else device
            self.global_step = 0
            self.num_episodes = 0
            self.seed = seed
            self.np_random = np.random.default_rng(self.seed)

        def extract_env_info(self, env) -> None:
            if env is None:
                return
            self.env = env
            osp, asp = env.observation_space, env.action_space
            if hasattr(osp, "n") and not getattr(osp, "shape", ()):
                self.observation_shape, self.observation_dim = (1,), osp.n
            else:
                self.observation_shape, self.observation_dim = tuple(osp.shape), osp.shape[0]
            self.action_space = asp
            if hasattr(asp, "n"):
                self.action_shape, self.action_dim = (1,), asp.n
            else:
                self.action_shape, self.action_dim = tuple(asp.shape), asp.shape[0]
            self.reward_dim = env.unwrapped.reward_space.shape[0]

        @abstractmethod
        def get_config(self) -> dict:
            ...

        def register_additional_config(self, conf=None) -> None:
            pass

        def setup_wandb(self, *a, **k) -> None:
            raise ImportError("wandb logging needs the reference package (morl_baselines) and wandb installed")

        def close_wandb(self) -> None:
            pass

    def _eval_mo(agent, env, w, scalarization=np.dot):
        """One greedy episode (``common/evaluation.py:23-67``)."""
        obs, _ = env.reset()
        done = False
        vec_return, disc_vec_return = np.zeros_like(w, dtype=np.float64), np.zeros_like(w, dtype=np.float64)
        gamma = 1.0
        while not done:
            obs, r, terminated, truncated, _ = env.step(agent.eval(obs, w))
            done = terminated or truncated
            vec_return = vec_return + r
            disc_vec_return = disc_vec_return + gamma * np.asarray(r)
            gamma *= agent.gamma
        if w is None:
            return float(vec_return), float(disc_vec_return), vec_return, disc_vec_return
        return scalarization(w, vec_return), scalarization(w, disc_vec_return), vec_return, disc_vec_return


def reference_method(module: str, cls: str, name: str):
    """The reference's own (unbound) method ``module.cls.name`` when ``morl_baselines`` is importable, else None.  The host mirrors
    hand their env-stepping loops to it -- ``Envelope.train`` is the reference's loop, unchanged, driving OUR ``act`` / ``update`` /
    replay buffer -- and keep their own restatement only for machines without the reference (the GPU test box)."""
    if not HAVE_REFERENCE_API:
        return None
    try:
        import importlib
        return getattr(getattr(importlib.import_module(module), cls), name)
    except Exception:
        return None
