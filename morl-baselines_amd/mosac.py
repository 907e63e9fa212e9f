"""MOSAC on the HIP actor-critic engine, with the reference's class surface
(``single_policy/ser/mosac_continuous_action.py``) -- the learner MORL/D runs for every sub-problem.

``update()`` is one ``morl_ac_update`` call: actor sample at s', twin target critics, scalarised TD target, twin
critics forward / backward, Adam, ``policy_freq`` actor iterations (each: loss through the updated critics, Adam, and
with ``autotune`` a fresh log-prob and one Adam step on ``log_alpha``), Polyak.  ``alpha`` lives on the device -- the
reference's per-iteration ``.item()`` disappears; reading ``agent.alpha`` synchronises on demand.

A ``MOSAC`` can own its engine (stand-alone use) or be handed a member slice of a population engine by ``MORLD``
(``engine=...``): its parameters, targets, Adam moments and step counters are then rows of the population's buffers and
``MORLD`` advances many members with one launch sequence.
"""
from __future__ import annotations

import time
from copy import deepcopy
from typing import Optional, Union

import numpy as np
import torch as th

from .ac_engine import ALGO_MOSAC, ACEngine
from .acnets import PolicyShell, SoftQShell, adam_state_dict, as_f32, bind, load_adam_state_dict, noise_device, randn
from .api import MOPolicy
from .native import NativeLib, load_library
from .replay import ReplayBuffer


class MOSAC(MOPolicy):
    """Multi-objective SAC with a multi-objective critic and weighted-sum scalarisation."""

    def __init__(self, env, weights: np.ndarray, scalarization=th.matmul, buffer_size: int = int(1e6),
                 gamma: float = 0.99, tau: float = 0.005, batch_size: int = 128, learning_starts: int = int(1e3),
                 net_arch=[256, 256], policy_lr: float = 3e-4, q_lr: float = 1e-3, policy_freq: int = 2,
                 target_net_freq: int = 1, alpha: float = 0.2, autotune: bool = True, id: Optional[int] = None,
                 device: Union[th.device, str] = "auto", log: bool = True, seed: int = 42,
                 parent_rng: Optional[np.random.Generator] = None, lib: Optional[NativeLib] = None,
                 engine: Optional[ACEngine] = None):
        super().__init__(id, device)
        if scalarization is not th.matmul:
            raise NotImplementedError("the HIP engine scalarises with the weighted sum (th.matmul) only")
        self.seed, self.parent_rng = seed, parent_rng
        self.np_random = parent_rng if parent_rng is not None else np.random.default_rng(self.seed)
        self.env = env
        if not (hasattr(env.action_space, "low") and hasattr(env.action_space, "high")):
            raise AssertionError("only continuous action space is supported")
        self.obs_shape = tuple(env.observation_space.shape)
        self.action_shape = tuple(env.action_space.shape)
        self.reward_dim = env.unwrapped.reward_space.shape[0]
        self.scalarization = scalarization
        self.batch_size, self.buffer_size, self.gamma, self.tau = batch_size, buffer_size, gamma, tau
        self.learning_starts, self.net_arch = learning_starts, net_arch
        self.policy_lr, self.q_lr, self.policy_freq, self.target_net_freq = policy_lr, q_lr, policy_freq, target_net_freq
        self.lib = lib or (engine.lib if engine is not None else load_library())
        D, Ad = int(np.prod(self.obs_shape)), int(np.prod(self.action_shape))
        low, high = np.asarray(env.action_space.low), np.asarray(env.action_space.high)
        self.engine = engine or ACEngine(ALGO_MOSAC, D, Ad, self.reward_dim, net_arch, action_low=low, action_high=high,
                                         max_rows=batch_size, device=self.device, lib=self.lib)
        e = self.engine
        self.set_weights(np.asarray(weights))
        # construction order (torch-RNG consumption) of mosac_continuous_action.py:213-251
        self.actor = PolicyShell(D, Ad, net_arch, ("fc_mean", "fc_logstd"), low, high)
        self.qf1, self.qf2 = SoftQShell(D + Ad, self.reward_dim, net_arch), SoftQShell(D + Ad, self.reward_dim, net_arch)
        self.qf1_target, self.qf2_target = (SoftQShell(D + Ad, self.reward_dim, net_arch) for _ in range(2))
        bind(self.actor, e.policy_views(e.pol))
        bind(self.qf1, e.q_views(e.q, 0, 0))
        bind(self.qf2, e.q_views(e.q, 0, 1))
        bind(self.qf1_target, e.q_views(e.q_target, 0, 0), copy_in=False)
        bind(self.qf2_target, e.q_views(e.q_target, 0, 1), copy_in=False)
        e.q_target.copy_(e.q)
        self.autotune = autotune
        if self.autotune:
            self.target_entropy = -float(np.prod(self.action_shape))
            e.log_alpha.zero_()
            self._alpha_const = None
        else:
            self.target_entropy = 0.0
            self._alpha_const = float(alpha)
        self._q_step = self._p_step = 0
        self.env.observation_space.dtype = np.float32
        self.buffer = ReplayBuffer(obs_shape=self.obs_shape, action_dim=self.action_shape[0], rew_dim=self.reward_dim,
                                   max_size=self.buffer_size, device=self.device, lib=self.lib)
        self.log = log
        self._out = None

    # -- entropy coefficient ----------------------------------------------------------------------------------------------
    @property
    def log_alpha(self) -> th.Tensor:
        return self.engine.log_alpha

    @property
    def alpha(self) -> float:
        """Host value (synchronises when the coefficient is learnt)."""
        if self._alpha_const is not None:
            return self._alpha_const
        return float(self.engine.log_alpha[0].exp().item())

    @alpha.setter
    def alpha(self, v: float) -> None:
        if self._alpha_const is not None:
            self._alpha_const = float(v)
        else:
            self.engine.log_alpha[0] = float(np.log(v))

    def get_config(self) -> dict:
        return {"env_id": self.env.unwrapped.spec.id, "buffer_size": self.buffer_size, "gamma": self.gamma,
                "tau": self.tau, "batch_size": self.batch_size, "learning_starts": self.learning_starts,
                "net_arch": self.net_arch, "policy_lr": self.policy_lr, "q_lr": self.q_lr,
                "policy_freq": self.policy_freq, "target_net_freq": self.target_net_freq, "alpha": self.alpha,
                "autotune": self.autotune, "seed": self.seed}

    def __deepcopy__(self, memo):
        """``mosac_continuous_action.py:298-340``: a new learner with copies of the networks, moments and buffer."""
        copied = type(self)(env=self.env, weights=self.weights, scalarization=self.scalarization,
                            buffer_size=self.buffer_size, gamma=self.gamma, tau=self.tau, batch_size=self.batch_size,
                            learning_starts=self.learning_starts, net_arch=self.net_arch, policy_lr=self.policy_lr,
                            q_lr=self.q_lr, policy_freq=self.policy_freq, target_net_freq=self.target_net_freq,
                            alpha=self.alpha, autotune=self.autotune, id=self.id, device=self.device, log=self.log,
                            seed=self.seed, parent_rng=self.parent_rng, lib=self.lib)
        src, dst = self.engine, copied.engine
        for name in ("q", "q_target", "pol", "log_alpha"):
            getattr(dst, name).copy_(getattr(src, name))
        # the reference re-creates the optimisers for the copy: fresh Adam state
        copied.global_step = self.global_step
        copied.buffer = deepcopy(self.buffer)
        return copied

    def get_buffer(self):
        return self.buffer

    def set_buffer(self, buffer):
        self.buffer = buffer

    def get_policy_net(self) -> th.nn.Module:
        return self.actor

    def set_weights(self, weights: np.ndarray):
        self.weights = weights
        self.weights_tensor = th.from_numpy(np.asarray(self.weights)).float().to(self.engine.q.device)

    # -- checkpoints (mosac_continuous_action.py:362-412) -----------------------------------------------------------------
    def _steps(self):
        e = self.engine
        if e.q_steps is not None:
            return int(e.q_steps[0].item()), int(e.pol_steps[0].item())
        return self._q_step, self._p_step

    def get_save_dict(self, save_replay_buffer: bool = False) -> dict:
        e = self.engine
        qs, ps = self._steps()
        qv = lambda buf: e.q_views(buf, 0, 0) + e.q_views(buf, 0, 1)  # noqa: E731
        d = {"actor_state_dict": self.actor.state_dict(), "qf1_state_dict": self.qf1.state_dict(),
             "qf2_state_dict": self.qf2.state_dict(), "qf1_target_state_dict": self.qf1_target.state_dict(),
             "qf2_target_state_dict": self.qf2_target.state_dict(),
             "actor_optimizer_state_dict": adam_state_dict(e.policy_views(e.pol), e.policy_views(e.pol_exp_avg),
                                                           e.policy_views(e.pol_exp_avg_sq), ps, self.policy_lr),
             "q_optimizer_state_dict": adam_state_dict(qv(e.q), qv(e.q_exp_avg), qv(e.q_exp_avg_sq), qs, self.q_lr),
             "weights": self.weights, "alpha": self.alpha}
        if save_replay_buffer:
            d["buffer"] = self.buffer
        if self.autotune:
            d["log_alpha"] = e.log_alpha.detach().clone()
            d["a_optimizer_state_dict"] = adam_state_dict([e.log_alpha], [e.log_alpha_exp_avg], [e.log_alpha_exp_avg_sq],
                                                          ps, self.q_lr)
        return d

    def load(self, save_dict: Optional[dict] = None, path: Optional[str] = None, load_replay_buffer: bool = True):
        if save_dict is None:
            assert path is not None, "Either save_dict or path should be provided."
            save_dict = th.load(path, map_location=self.device, weights_only=False)
        e = self.engine
        qv = lambda buf: e.q_views(buf, 0, 0) + e.q_views(buf, 0, 1)  # noqa: E731
        self.actor.load_state_dict(save_dict["actor_state_dict"])
        self.qf1.load_state_dict(save_dict["qf1_state_dict"])
        self.qf2.load_state_dict(save_dict["qf2_state_dict"])
        self.qf1_target.load_state_dict(save_dict["qf1_target_state_dict"])
        self.qf2_target.load_state_dict(save_dict["qf2_target_state_dict"])
        ps = load_adam_state_dict(save_dict["actor_optimizer_state_dict"], e.policy_views(e.pol_exp_avg),
                                  e.policy_views(e.pol_exp_avg_sq))
        qs = load_adam_state_dict(save_dict["q_optimizer_state_dict"], qv(e.q_exp_avg), qv(e.q_exp_avg_sq))
        if "log_alpha" in save_dict:
            e.log_alpha.copy_(save_dict["log_alpha"].to(e.log_alpha.device).reshape(-1))
            load_adam_state_dict(save_dict["a_optimizer_state_dict"], [e.log_alpha_exp_avg], [e.log_alpha_exp_avg_sq])
        self._q_step, self._p_step = qs, ps
        if e.q_steps is not None:
            e.q_steps[0], e.pol_steps[0] = qs, ps
        if load_replay_buffer:
            self.buffer = save_dict["buffer"]
        self.set_weights(save_dict["weights"])
        if not self.autotune:
            self._alpha_const = float(save_dict["alpha"])

    # -- acting ------------------------------------------------------------------------------------------------------------
    @th.no_grad()
    def _sampled_action(self, obs) -> np.ndarray:
        e = self.engine
        obs = as_f32(np.asarray(obs, dtype=np.float32), e.q.device).reshape(1, -1)
        eps = randn((1, e.Ad), e.q.device)
        return e.policy_forward(obs, eps=eps)[0, 0].detach().cpu().numpy()

    def eval(self, obs: np.ndarray, w: Optional[np.ndarray] = None) -> Union[int, np.ndarray]:
        """``mosac_continuous_action.py:414-428``: the reference evaluates with a SAMPLED action."""
        return self._sampled_action(obs)

    # -- the hot path (mosac_continuous_action.py:430-489) ----------------------------------------------------------------
    def update_inputs(self):
        """Sample this learner's batch and noise (what ``update`` feeds the engine); ``MORLD`` stacks these."""
        mb_obs, mb_act, mb_rewards, mb_next_obs, mb_dones, _ = self.buffer.sample(self.batch_size, to_tensor=True,
                                                                                  device=self.device)
        return mb_obs, mb_act, mb_rewards, mb_next_obs, mb_dones.reshape(-1)

    def make_cfg(self):
        e = self.engine
        do_policy = self.global_step % self.policy_freq == 0
        return e.make_cfg(gamma=self.gamma, tau=self.tau, alpha=self._alpha_const or 0.0, q_lr=self.q_lr,
                          policy_lr=self.policy_lr, alpha_lr=self.q_lr, q_step=self._q_step + 1,
                          policy_step=self._p_step + 1, do_policy=do_policy, policy_iters=self.policy_freq,
                          do_target=(self.global_step % self.target_net_freq == 0), autotune=self.autotune,
                          target_entropy=self.target_entropy)

    def update(self):
        e = self.engine
        obs, act, rew, nobs, dones = self.update_inputs()
        B, Ad = obs.shape[0], e.Ad
        cfg = self.make_cfg()
        # draws in the reference's order (next action; per actor iteration: pi, then the alpha re-sample), one call each,
        # so that the CPU test backend consumes torch's generator exactly as mosac_continuous_action.py:436-468 does
        pf = self.policy_freq
        eps = th.empty((1 + 2 * pf, B, Ad), dtype=th.float32, device=noise_device(e.q.device))
        eps[0].normal_()
        if cfg.do_policy:
            for k in range(pf):
                eps[1 + k].normal_()
                if self.autotune:
                    eps[1 + pf + k].normal_()
        eps = eps.to(e.q.device)
        self._out = e.update(cfg, obs=obs, actions=act, rewards=rew, next_obs=nobs, dones=dones, w=self.weights_tensor,
                             eps_next=eps[0], eps_pi=eps[1:1 + self.policy_freq], eps_alpha=eps[1 + self.policy_freq:],
                             want=("critic_loss", "q_losses", "policy_loss", "alpha_loss"))
        self.note_update(bool(cfg.do_policy))
        if self.global_step % 100 == 0 and self.log:
            import wandb
            s = f"_{self.id}" if self.id is not None else ""
            to_log = {f"losses{s}/alpha": self.alpha, f"losses{s}/qf1_loss": float(self._out["q_losses"][0, 0].item()),
                      f"losses{s}/qf2_loss": float(self._out["q_losses"][0, 1].item()),
                      f"losses{s}/qf_loss": float(self._out["critic_loss"][0].item()) / 2.0,
                      f"losses{s}/actor_loss": float(self._out["policy_loss"][0].item()),
                      "global_step": self.global_step}
            if self.autotune:
                to_log[f"losses{s}/alpha_loss"] = float(self._out["alpha_loss"][0].item())
            wandb.log(to_log)

    def note_update(self, did_policy: bool) -> None:
        """Host mirror of the optimiser step counters (the device counters, if any, advance inside the call)."""
        self._q_step += 1
        if did_policy:
            self._p_step += self.policy_freq

    def train(self, total_timesteps: int, eval_env=None, start_time=None):
        """``mosac_continuous_action.py:508-571``."""
        if start_time is None:
            start_time = time.time()
        obs, _ = self.env.reset()
        for step in range(total_timesteps):
            if self.global_step < self.learning_starts:
                actions = self.env.action_space.sample()
            else:
                actions = self._sampled_action(obs)
            next_obs, rewards, terminated, truncated, infos = self.env.step(actions)
            real_next_obs = next_obs
            if "final_observation" in infos:
                real_next_obs = infos["final_observation"]
            self.buffer.add(obs=obs, next_obs=real_next_obs, action=actions, reward=rewards, done=terminated)
            obs = next_obs
            if terminated or truncated:
                obs, _ = self.env.reset()
            if self.global_step > self.learning_starts:
                self.update()
                if self.log and self.global_step % 100 == 0:
                    import wandb
                    wandb.log({"charts/SPS": int(self.global_step / (time.time() - start_time)),
                               "global_step": self.global_step})
            self.global_step += 1
