"""MOSAC with discrete actions on the HIP actor-critic engine (``single_policy/ser/mosac_discrete_action.py``) -- MORL/D's
alternative sub-problem learner.

``update()`` is one ``morl_ac_update`` call (algo SACD): actor logits at s', twin target critics Q(s', .), the exact
expectation over actions of ``min_n(Q_n . w) - alpha log pi`` as the soft target, twin critics forward / backward / Adam
(eps 1e-4 like the reference), actor loss through the updated critics with its analytic logit gradient, Adam, the
entropy-coefficient step from the same probabilities, Polyak.  Acting samples from ``Categorical(logits)`` on the host
exactly as the reference does (the logits come from the device).
"""
from __future__ import annotations

import time
from copy import deepcopy
from typing import Optional, Union

import numpy as np
import torch as th
from torch import nn
from torch.distributions import Categorical

from .ac_engine import ALGO_SACD, ACEngine
from .acnets import adam_state_dict, as_f32, bind, build_mlp, layer_init, load_adam_state_dict, noise_device
from .api import MOPolicy
from .native import NativeLib, load_library
from .replay import ReplayBuffer


class _DiscreteNetShell(nn.Module):
    """``feature_extractor`` (first hidden layer) + ``net`` (the rest): the parameter order of one plain MLP."""

    def __init__(self, obs_dim, out_dim, net_arch):
        super().__init__()
        self.feature_extractor = build_mlp(obs_dim, -1, net_arch[:1])
        self.net = build_mlp(net_arch[0], out_dim, net_arch[1:])
        self.apply(layer_init)


class MOSACDiscrete(MOPolicy):
    """Multi-objective SAC for discrete action spaces (weighted-sum scalarisation)."""

    def __init__(self, env, weights: np.ndarray, scalarization=th.matmul, buffer_size: int = int(1e6),
                 gamma: float = 0.99, tau: float = 1.0, batch_size: int = 128, learning_starts: int = int(2e4),
                 net_arch=[256, 256], policy_lr: float = 3e-4, q_lr: float = 3e-4, update_frequency: int = 4,
                 target_net_freq: int = 2000, alpha: float = 0.2, autotune: bool = True,
                 target_entropy_scale: float = 0.89, id: Optional[int] = None, device: Union[th.device, str] = "auto",
                 log: bool = True, seed: int = 42, parent_rng: Optional[np.random.Generator] = None,
                 lib: Optional[NativeLib] = None, engine: Optional[ACEngine] = None):
        super().__init__(id, device)
        if scalarization is not th.matmul:
            raise NotImplementedError("the HIP engine scalarises with the weighted sum (th.matmul) only")
        if len(net_arch) < 2:
            raise ValueError("net_arch needs at least two entries (feature extractor + net), as in the reference")
        self.seed, self.parent_rng = seed, parent_rng
        self.np_random = parent_rng if parent_rng is not None else np.random.default_rng(self.seed)
        self.env = env
        assert hasattr(env.action_space, "n"), "only discrete action space is supported"
        self.obs_shape = tuple(env.observation_space.shape)
        self.action_dim = int(env.action_space.n)
        self.reward_dim = env.unwrapped.reward_space.shape[0]
        self.scalarization = scalarization
        self.batch_size, self.buffer_size, self.gamma, self.tau = batch_size, buffer_size, gamma, tau
        self.learning_starts, self.net_arch = learning_starts, net_arch
        self.policy_lr, self.q_lr = policy_lr, q_lr
        self.update_frequency, self.target_net_freq = update_frequency, target_net_freq
        assert self.target_net_freq % self.update_frequency == 0, "target_net_freq should be divisible by update_frequency"
        self.target_entropy_scale = target_entropy_scale
        self.lib = lib or (engine.lib if engine is not None else load_library())
        D, A, R = int(np.prod(self.obs_shape)), self.action_dim, self.reward_dim
        self.engine = engine or ACEngine(ALGO_SACD, D, A, R, net_arch, action_low=0.0, action_high=1.0,
                                         max_rows=batch_size, device=self.device, lib=self.lib)
        e = self.engine
        self.set_weights(np.asarray(weights))
        # construction order (torch-RNG consumption) of mosac_discrete_action.py:206-243
        self.actor = _DiscreteNetShell(D, A, net_arch)
        self.qf1, self.qf2 = _DiscreteNetShell(D, A * R, net_arch), _DiscreteNetShell(D, A * R, net_arch)
        self.qf1_target, self.qf2_target = (_DiscreteNetShell(D, A * R, net_arch) for _ in range(2))
        bind(self.actor, e.policy_views(e.pol))
        bind(self.qf1, e.q_views(e.q, 0, 0))
        bind(self.qf2, e.q_views(e.q, 0, 1))
        bind(self.qf1_target, e.q_views(e.q_target, 0, 0), copy_in=False)
        bind(self.qf2_target, e.q_views(e.q_target, 0, 1), copy_in=False)
        e.q_target.copy_(e.q)
        self.autotune = autotune
        if self.autotune:
            self.target_entropy = float(-self.target_entropy_scale * th.log(1 / th.tensor(self.action_dim)))
            e.log_alpha.zero_()
            self._alpha_const = None
        else:
            self.target_entropy = 0.0
            self._alpha_const = float(alpha)
        self._q_step = self._p_step = 0
        self.env.observation_space.dtype = np.float32
        self.buffer = ReplayBuffer(obs_shape=self.obs_shape, action_dim=1, rew_dim=self.reward_dim,
                                   max_size=self.buffer_size, device=self.device, lib=self.lib)
        self.log = log
        self._out = None

    @property
    def log_alpha(self) -> th.Tensor:
        return self.engine.log_alpha

    @property
    def alpha(self) -> float:
        if self._alpha_const is not None:
            return self._alpha_const
        return float(self.engine.log_alpha[0].exp().item())

    def get_config(self) -> dict:
        return {"env_id": self.env.unwrapped.spec.id, "buffer_size": self.buffer_size, "gamma": self.gamma,
                "tau": self.tau, "batch_size": self.batch_size, "learning_starts": self.learning_starts,
                "net_arch": self.net_arch, "policy_lr": self.policy_lr, "q_lr": self.q_lr,
                "update_frequency": self.update_frequency, "target_net_freq": self.target_net_freq, "alpha": self.alpha,
                "autotune": self.autotune, "target_entropy_scale": self.target_entropy_scale, "seed": self.seed}

    def __deepcopy__(self, memo):
        copied = type(self)(env=self.env, weights=self.weights, scalarization=self.scalarization,
                            buffer_size=self.buffer_size, gamma=self.gamma, tau=self.tau, batch_size=self.batch_size,
                            learning_starts=self.learning_starts, net_arch=self.net_arch, policy_lr=self.policy_lr,
                            q_lr=self.q_lr, update_frequency=self.update_frequency,
                            target_net_freq=self.target_net_freq, alpha=self.alpha, autotune=self.autotune,
                            target_entropy_scale=self.target_entropy_scale, id=self.id, device=self.device, log=self.log,
                            seed=self.seed, parent_rng=self.parent_rng, lib=self.lib)
        for name in ("q", "q_target", "pol", "log_alpha"):
            getattr(copied.engine, name).copy_(getattr(self.engine, name))
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

    def _steps(self):
        e = self.engine
        if e.q_steps is not None:
            return int(e.q_steps[0].item()), int(e.pol_steps[0].item())
        return self._q_step, self._p_step

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

    # -- acting: Categorical(logits).sample() on the host, as the reference (mosac_discrete_action.py:98-106) ---------------
    @th.no_grad()
    def _sample_action(self, obs) -> np.ndarray:
        e = self.engine
        o = as_f32(np.asarray(obs, dtype=np.float32), e.q.device).reshape(1, -1)
        logits = e.policy_forward(o)[0]                       # (1, A)
        action = Categorical(logits=logits.to(noise_device(logits.device))).sample()   # (see acnets.HOST_NOISE)
        return action[0].detach().cpu().numpy()

    def eval(self, obs: np.ndarray, w: Optional[np.ndarray] = None) -> Union[int, np.ndarray]:
        return self._sample_action(obs)

    # -- the hot path (mosac_discrete_action.py:440-503) ---------------------------------------------------------------------
    def update_inputs(self):
        mb_obs, mb_act, mb_rewards, mb_next_obs, mb_dones, _ = self.buffer.sample(self.batch_size, to_tensor=True,
                                                                                  device=self.device)
        return mb_obs, mb_act.reshape(-1), mb_rewards, mb_next_obs, mb_dones.reshape(-1)

    def make_cfg(self):
        return self.engine.make_cfg(gamma=self.gamma, tau=self.tau, alpha=self._alpha_const or 0.0, q_lr=self.q_lr,
                                    policy_lr=self.policy_lr, alpha_lr=self.q_lr, q_step=self._q_step + 1,
                                    policy_step=self._p_step + 1,
                                    do_target=(self.global_step % self.target_net_freq == 0), autotune=self.autotune,
                                    target_entropy=self.target_entropy, eps=1e-4)

    def update(self):
        obs, act, rew, nobs, dones = self.update_inputs()
        cfg = self.make_cfg()
        # The reference's update() calls actor.get_action() twice and throws the sampled actions away
        # (mosac_discrete_action.py:450, :476): those two Categorical.sample() calls still advance torch's generator.  Burn
        # the same draws (one per row, independent of the probabilities) so that seeded runs keep acting identically.
        burn = th.ones((obs.shape[0], self.action_dim), dtype=th.float32, device=noise_device(self.engine.q.device))
        th.multinomial(burn, 1, True)
        th.multinomial(burn, 1, True)
        self._out = self.engine.update(cfg, obs=obs, actions=act, rewards=rew, next_obs=nobs, dones=dones,
                                       w=self.weights_tensor, want=("critic_loss", "q_losses", "policy_loss", "alpha_loss"))
        self.note_update(True)
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
        self._q_step += 1
        self._p_step += 1

    def train(self, total_timesteps: int, eval_env=None, start_time=None, verbose: bool = False):
        """``mosac_discrete_action.py:529-603``."""
        if start_time is None:
            start_time = time.time()
        obs, _ = self.env.reset()
        for _ in range(total_timesteps):
            if self.global_step < self.learning_starts:
                actions = self.env.action_space.sample()
            else:
                actions = self._sample_action(obs)
            next_obs, rewards, terminated, truncated, infos = self.env.step(actions)
            real_next_obs = next_obs
            if "final_observation" in infos:
                real_next_obs = infos["final_observation"]
            self.buffer.add(obs=obs, next_obs=real_next_obs, action=actions, reward=rewards, done=terminated)
            obs = next_obs
            if terminated or truncated:
                obs, _ = self.env.reset()
            if self.global_step > self.learning_starts:
                if self.global_step % self.update_frequency == 0:
                    self.update()
                if self.log and self.global_step % 100 == 0:
                    import wandb
                    wandb.log({"charts/SPS": int(self.global_step / (time.time() - start_time)),
                               "global_step": self.global_step})
            self.global_step += 1
