"""GPI-PD / GPI-LS with discrete actions on the HIP engine (``multi_policy/gpi_pd/gpi_pd.py``), model-free path.

Same constructor arguments and methods as the reference's ``GPIPD`` (``update``, ``eval``, ``gpi_action``,
``max_action``, ``set_weight_support``, ``train_iteration``, ``train``, ``save`` / ``load``, ``get_config``).  On the
device: the prioritised replay (sum tree + gather), the ensemble of conditioned Q-nets and every gradient update as ONE
``morl_gpi_update`` call (min-over-ensemble TD target, GPI envelope target over the sampled weights, Huber loss,
backward, per-net clipping, Adam, PER errors), the GPI action as one ``morl_gpi_action`` call, and ``_reset_priorities``
as chunked ``morl_gpi_priorities`` calls over the device-resident records.

``dyna=True`` (GPI-PD proper): the probabilistic dynamics ensemble is trained on the device (``dynamics.py``,
``morl_ens_*``), the imagined rollouts pick their GPI actions for the whole observation batch in one
``morl_gpi_actions`` call and enter the model buffer with one batched ``add`` instead of the reference's per-row Python
loop (gpi_pd.py:394-397).  Dropout is not applied when acting, rolling out or re-prioritising (the reference leaves the
nets in train mode there).
"""
from __future__ import annotations

import os
import random
from typing import Callable, List, Optional, Union

import numpy as np
import torch as th
from torch import nn

from .evaluation import front_returns
from .acnets import adam_state_dict, bind, build_mlp, layer_init, load_adam_state_dict
from .api import MOAgent, MOPolicy
from .envelope import linearly_decaying_value
from .gpi_engine import GPIEngine
from .gpi_pd_continuous import unique_tol
from .native import NativeLib, load_library
from .replay import PrioritizedReplayBuffer, ReplayBuffer


class QNetShell(nn.Module):
    """Parameter shell of ``QNet`` (gpi_pd.py:41-76): ``weights_features``, ``state_features``, ``net``."""

    def __init__(self, obs_dim, action_dim, rew_dim, net_arch, drop_rate=0.01, layer_norm=True):
        super().__init__()
        self.weights_features = build_mlp(rew_dim, -1, net_arch[:1])
        self.state_features = build_mlp(obs_dim, -1, net_arch[:1])
        self.net = build_mlp(net_arch[0], action_dim * rew_dim, net_arch[1:], drop_rate=drop_rate, layer_norm=layer_norm)
        self.apply(layer_init)


class GPIPD(MOPolicy, MOAgent):
    """GPI-PD (Alegre et al., AAMAS 2023) -- model-free path on the MI355X."""

    def __init__(self, env, learning_rate: float = 3e-4, initial_epsilon: float = 0.01, final_epsilon: float = 0.01,
                 epsilon_decay_steps: int = None, tau: float = 1.0, target_net_update_freq: int = 1000,
                 buffer_size: int = int(1e6), net_arch: List = [256, 256, 256, 256], num_nets: int = 2,
                 batch_size: int = 128, learning_starts: int = 100, gradient_updates: int = 20, gamma: float = 0.99,
                 max_grad_norm: Optional[float] = None, use_gpi: bool = True, dyna: bool = True, per: bool = True,
                 gpi_pd: bool = True, alpha_per: float = 0.6, min_priority: float = 0.01, drop_rate: float = 0.01,
                 layer_norm: bool = True, dynamics_normalize_inputs: bool = False,
                 dynamics_uncertainty_threshold: float = 1.5, dynamics_train_freq: Callable = lambda timestep: 250,
                 dynamics_rollout_len: int = 1, dynamics_rollout_starts: int = 5000, dynamics_rollout_freq: int = 250,
                 dynamics_rollout_batch_size: int = 25000, dynamics_buffer_size: int = 100000,
                 dynamics_net_arch: List = [256, 256, 256], dynamics_ensemble_size: int = 5,
                 dynamics_num_elites: int = 2, real_ratio: float = 0.5, project_name: str = "MORL-Baselines",
                 experiment_name: str = "GPI-PD", wandb_entity: Optional[str] = None, log: bool = True,
                 seed: Optional[int] = None, device: Union[th.device, str] = "auto", lib: Optional[NativeLib] = None,
                 max_support: int = 64, dynamics_max_rows: int = 10000):
        MOAgent.__init__(self, env, device=device, seed=seed)
        MOPolicy.__init__(self, device=device)
        self.learning_rate, self.initial_epsilon, self.epsilon = learning_rate, initial_epsilon, initial_epsilon
        self.epsilon_decay_steps, self.final_epsilon, self.tau = epsilon_decay_steps, final_epsilon, tau
        self.target_net_update_freq, self.gamma, self.max_grad_norm = target_net_update_freq, gamma, max_grad_norm
        self.use_gpi, self.buffer_size, self.net_arch = use_gpi, buffer_size, net_arch
        self.learning_starts, self.batch_size, self.gradient_updates = learning_starts, batch_size, gradient_updates
        self.num_nets, self.drop_rate, self.layer_norm = num_nets, drop_rate, layer_norm
        self.lib = lib or load_library()
        self.engine = GPIEngine(self.observation_dim, self.action_dim, self.reward_dim, net_arch, max_rows=2 * batch_size,
                                max_support=max_support, num_nets=num_nets, layer_norm=layer_norm, drop_rate=drop_rate,
                                device=self.device, lib=self.lib)
        e = self.engine
        mk = lambda: QNetShell(self.observation_dim, self.action_dim, self.reward_dim, net_arch, drop_rate, layer_norm)  # noqa: E731
        self.q_nets = [mk() for _ in range(num_nets)]
        self.target_q_nets = [mk() for _ in range(num_nets)]
        for n in range(num_nets):
            bind(self.q_nets[n], e.views(e.q, n))
            bind(self.target_q_nets[n], e.views(e.q_target, n), copy_in=False)
        e.q_target.copy_(e.q)
        self.per, self.gpi_pd = per, gpi_pd
        buf_cls = PrioritizedReplayBuffer if per else ReplayBuffer
        self.replay_buffer = buf_cls(self.observation_shape, 1, rew_dim=self.reward_dim, max_size=buffer_size,
                                     action_dtype=np.uint8, device=self.device, lib=self.lib)
        self.min_priority, self.alpha = min_priority, alpha_per
        # model-based part (gpi_pd.py:232-266)
        self.dyna, self.dynamics_net_arch, self.dynamics, self.dynamics_buffer = dyna, dynamics_net_arch, None, None
        if self.dyna:
            from .dynamics import ProbabilisticEnsemble
            self.dynamics = ProbabilisticEnsemble(input_dim=self.observation_dim + self.action_dim,
                                                  output_dim=self.observation_dim + self.reward_dim,
                                                  arch=self.dynamics_net_arch, normalize_inputs=dynamics_normalize_inputs,
                                                  ensemble_size=dynamics_ensemble_size, num_elites=dynamics_num_elites,
                                                  device=self.device, lib=self.lib,
                                                  max_rows=dynamics_max_rows)   # >= fit batch (256) / holdout / rollout rows
            self.dynamics_buffer = ReplayBuffer(self.observation_shape, 1, rew_dim=self.reward_dim,
                                                max_size=dynamics_buffer_size, action_dtype=np.uint8, device=self.device,
                                                lib=self.lib)
        self.dynamics_train_freq, self.dynamics_buffer_size = dynamics_train_freq, dynamics_buffer_size
        self.dynamics_normalize_inputs, self.dynamics_num_elites = dynamics_normalize_inputs, dynamics_num_elites
        self.dynamics_ensemble_size, self.dynamics_rollout_len = dynamics_ensemble_size, dynamics_rollout_len
        self.dynamics_rollout_freq, self.dynamics_rollout_batch_size = dynamics_rollout_freq, dynamics_rollout_batch_size
        self.dynamics_uncertainty_threshold, self.real_ratio = dynamics_uncertainty_threshold, real_ratio
        self.dynamics_fit_kwargs = {}                 # forwarded to ProbabilisticEnsemble.fit (reference: defaults)
        self.model_termination_func = None            # optional override of the environment's termination rule
        self.dynamics_rollout_starts = dynamics_rollout_starts
        self.weight_support: List[th.Tensor] = []
        self.stacked_weight_support = None
        self.police_indices = []
        self._adam_step = 0
        self._drop_seed = (0 if seed is None else int(seed)) * 1000003 + 12345   # never touches self.np_random
        self._out = None
        self.experiment_name = experiment_name
        self.log = log
        if self.log:
            self.setup_wandb(project_name, experiment_name, wandb_entity)

    def get_config(self):
        return {"env_id": self.env.unwrapped.spec.id, "learning_rate": self.learning_rate,
                "initial_epsilon": self.initial_epsilon, "epsilon_decay_steps:": self.epsilon_decay_steps,
                "batch_size": self.batch_size, "per": self.per, "gpi_pd": self.gpi_pd, "alpha_per": self.alpha,
                "min_priority": self.min_priority, "tau": self.tau, "num_nets": self.num_nets,
                "clip_grand_norm": self.max_grad_norm, "target_net_update_freq": self.target_net_update_freq,
                "gamma": self.gamma, "net_arch": self.net_arch, "gradient_updates": self.gradient_updates,
                "buffer_size": self.buffer_size, "learning_starts": self.learning_starts, "dyna": self.dyna,
                "drop_rate": self.drop_rate, "layer_norm": self.layer_norm, "seed": self.seed}

    # -- checkpoints (gpi_pd.py:314-341) -------------------------------------------------------------------------------------
    def _all_views(self, buf):
        return [v for n in range(self.num_nets) for v in self.engine.views(buf, n)]

    def save(self, save_replay_buffer=True, save_dir="weights/", filename=None):
        if not os.path.isdir(save_dir):
            os.makedirs(save_dir)
        e = self.engine
        saved = {f"psi_net_{i}_state_dict": q.state_dict() for i, q in enumerate(self.q_nets)}
        saved["psi_nets_optimizer_state_dict"] = adam_state_dict(self._all_views(e.q), self._all_views(e.exp_avg),
                                                                 self._all_views(e.exp_avg_sq), self._adam_step,
                                                                 self.learning_rate)
        saved["M"] = self.weight_support
        if self.dyna:                                   # gpi_pd.py:323-324
            saved["dynamics_state_dict"] = self.dynamics.state_dict()
        if save_replay_buffer:
            saved["replay_buffer"] = self.replay_buffer
        filename = self.experiment_name if filename is None else filename
        th.save(saved, save_dir + "/" + filename + ".tar")

    def load(self, path, load_replay_buffer=True):
        params = th.load(path, map_location=self.device, weights_only=False)
        e = self.engine
        for i, (q, tq) in enumerate(zip(self.q_nets, self.target_q_nets)):
            q.load_state_dict(params[f"psi_net_{i}_state_dict"])
            tq.load_state_dict(params[f"psi_net_{i}_state_dict"])
        self._adam_step = load_adam_state_dict(params["psi_nets_optimizer_state_dict"], self._all_views(e.exp_avg),
                                               self._all_views(e.exp_avg_sq))
        self.set_weight_support([w.cpu().numpy() for w in params["M"]])
        if self.dyna and "dynamics_state_dict" in params:   # gpi_pd.py:338-339
            self.dynamics.load_state_dict(params["dynamics_state_dict"])
        if load_replay_buffer and "replay_buffer" in params:
            self.replay_buffer = params["replay_buffer"]

    def _sample_batch_experiences(self):
        """``gpi_pd.py:343-365``: real transitions, or a real / imagined mix once the model buffer is in use."""
        if not self.dyna or self.global_step < self.dynamics_rollout_starts or len(self.dynamics_buffer) == 0:
            return self.replay_buffer.sample(self.batch_size, to_tensor=True, device=self.device)
        num_real = int(self.batch_size * self.real_ratio)
        real = self.replay_buffer.sample(num_real, to_tensor=True, device=self.device)
        model = self.dynamics_buffer.sample(self.batch_size - num_real, to_tensor=True, device=self.device)
        mixed = tuple(th.cat([r.reshape(r.shape[0], -1), m.reshape(m.shape[0], -1)], dim=0) for r, m in zip(real[:5], model[:5]))
        return mixed + ((real[5],) if self.per else ())

    @th.no_grad()
    def _rollout_dynamics(self, w: th.Tensor):
        """``gpi_pd.py:367-414``: imagined transitions from GPI actions under the current weight."""
        from .dynamics import ModelEnv
        num_times = int(np.ceil(self.dynamics_rollout_batch_size / 10000))
        batch_size = min(self.dynamics_rollout_batch_size, 10000)
        added = 0
        w = th.as_tensor(w).to(self.engine.q.device, th.float32).reshape(-1)
        for _ in range(num_times):
            obs = self.replay_buffer.sample_obs(batch_size, to_tensor=False)
            model_env = ModelEnv(self.dynamics, self.env.unwrapped.spec.id, rew_dim=len(w),
                                 termination_func=self.model_termination_func)
            for h in range(self.dynamics_rollout_len):
                obs_t = th.as_tensor(np.ascontiguousarray(obs, dtype=np.float32)).to(self.engine.q.device)
                actions = self.engine.actions_batch(obs_t, w, self.stacked_weight_support).long()
                one_hot = th.nn.functional.one_hot(actions, num_classes=self.action_dim)
                next_obs_pred, r_pred, dones, info = model_env.step(obs_t, one_hot, deterministic=False)
                unc = info["uncertainty"]
                keep = unc < self.dynamics_uncertainty_threshold
                obs_h, act_h = obs_t.cpu().numpy(), actions.cpu().numpy()
                self.dynamics_buffer.add_batch(obs_h[keep], act_h[keep], r_pred[keep], next_obs_pred[keep], dones[keep])
                added += int(keep.sum())
                nonterm = ~dones.squeeze(-1)
                if nonterm.sum() == 0:
                    break
                obs = next_obs_pred[nonterm]
        self._last_rollout = {"imagined": added, "uncertainty_mean": float(unc.mean())}

    # -- the hot path (gpi_pd.py:416-562) --------------------------------------------------------------------------------------
    per_one_entry_enabled = True      # False: one sample / update / update_priorities entry per iteration (the A/B of the tests)

    def update(self, weight: th.Tensor):
        e = self.engine
        dev = e.q.device
        weight = th.as_tensor(weight).to(dev, th.float32).reshape(-1)
        critic_losses, priority, gpriority, deferred = [], None, None, []
        n_updates = self.gradient_updates if self.global_step >= self.dynamics_rollout_starts else 1
        real_only = not self.dyna or self.global_step < self.dynamics_rollout_starts or len(self.dynamics_buffer) == 0
        # (one tree-update launch holds ST_MAX_B = 1 024 entries: larger batches keep the per-iteration rounds, whose
        # update_priorities splits the update into blocks)
        per_one_entry = self.per_one_entry_enabled and self.per and real_only and self.replay_buffer._int_actions and \
            self.replay_buffer._Ad == 1 and self.batch_size <= self.replay_buffer.TREE_BLOCK
        B = self.batch_size
        doubled = len(self.weight_support) > 1
        if per_one_entry:
            # the whole loop as ONE library entry (morl_gpi_update_n_per): the host draws what the reference's loop draws, in its
            # order per generator (numpy: the B unit uniforms of each PrioritizedReplayBuffer.sample; random: choices / sample) --
            # none of it depends on what the device computes -- and the device samples, updates and re-prioritises per iteration
            rows = 2 * B if doubled else B
            sc = self.__dict__.get("_per_scratch")
            if sc is None or sc[0].shape[0] != rows:
                D, R = self.replay_buffer._D, self.replay_buffer._R
                sc = (th.empty((rows, D), dtype=th.float32, device=dev), th.empty((rows,), dtype=th.int32, device=dev),
                      th.empty((rows, R), dtype=th.float32, device=dev), th.empty((rows, D), dtype=th.float32, device=dev),
                      th.empty((rows,), dtype=th.float32, device=dev))
                self._per_scratch = sc
            u01 = np.empty((n_updates, B), dtype=np.float64)
        for g in range(n_updates):
            if per_one_entry:
                u01[g] = np.random.random_sample(B)
                s_obs, s_actions, s_rewards, s_next_obs, s_dones = sc
                idxes, n_per = True, B
            else:
                batch = self._sample_batch_experiences()
                s_obs, s_actions, s_rewards, s_next_obs, s_dones = batch[:5]
                idxes = batch[5] if self.per else None
                B = s_obs.size(0)
                n_per = idxes.numel() if idxes is not None else B      # the imagined rows of a Dyna batch carry no priority
            if doubled:
                if not per_one_entry:
                    s_obs, s_rewards, s_next_obs, s_dones = (x.reshape(B, -1).repeat(2, 1) for x in
                                                             (s_obs, s_rewards, s_next_obs, s_dones))
                    s_actions = s_actions.reshape(-1).repeat(2)
                w = th.vstack([weight.expand(B, -1)] + random.choices(self.weight_support, k=B))
            else:
                w = weight.repeat(B, 1)
            if len(self.weight_support) > 5:
                sampled_w = th.stack([weight] + random.sample(self.weight_support, k=4))
            else:
                sampled_w = th.stack(self.weight_support)
            self._adam_step += 1
            self._drop_seed += 1
            want = ("critic_loss",) + (("td_error",) if self.per else ()) + (("gtd_error",) if self.gpi_pd else ()) + \
                (("grad_norm",) if self.max_grad_norm is not None else ())
            kw = dict(obs=s_obs, actions=s_actions, rewards=s_rewards, next_obs=s_next_obs, dones=s_dones, w=w,
                      sampled_w=sampled_w, gamma=self.gamma, lr=self.learning_rate, adam_step=self._adam_step,
                      min_priority=self.min_priority, max_grad_norm=self.max_grad_norm, gpi_pd=self.gpi_pd,
                      n_per=(n_per if (self.per or self.gpi_pd) else 0), dropout_seed=self._drop_seed, want=want)
            if idxes is None or per_one_entry:
                # no prioritised replay: iteration g + 1 does not sample through what iteration g wrote, so the whole loop is
                # drawn first and submitted as ONE library entry below (morl_gpi_update_n)
                deferred.append(kw)
                continue
            out = e.update(**kw)
            self._out = out
            critic_losses.append(out["critic_loss"])
            if self.gpi_pd:
                gpriority = out["gtd_error"].clamp(min=self.min_priority).pow(self.alpha)
            if self.per:
                priority = out["td_error"].clamp(min=self.min_priority).pow(self.alpha)
            self.replay_buffer.update_priorities(idxes, gpriority if self.gpi_pd else priority)
        if deferred:
            if per_one_entry:
                outs, self._last_per_idx = e.update_n_per(deferred, buffer=self.replay_buffer, u01=u01, doubled=doubled,
                                                          use_gtd=self.gpi_pd, alpha=self.alpha, min_priority=self.min_priority)
                priority = outs[-1]["td_error"].clamp(min=self.min_priority).pow(self.alpha)
            else:
                outs = [e.update(**deferred[0])] if len(deferred) == 1 else e.update_n(deferred)
            self._out = outs[-1]
            critic_losses += [o["critic_loss"] for o in outs]
            if self.gpi_pd:
                gpriority = outs[-1]["gtd_error"].clamp(min=self.min_priority).pow(self.alpha)
        if self.tau != 1 or self.global_step % self.target_net_update_freq == 0:
            from . import ops
            ops.polyak(self.lib, e.q.view(-1), e.q_target.view(-1), self.tau)
        if self.epsilon_decay_steps is not None:
            self.epsilon = linearly_decaying_value(self.initial_epsilon, self.epsilon_decay_steps, self.global_step,
                                                   self.learning_starts, self.final_epsilon)
        if self.log and self.global_step % 100 == 0:
            import wandb
            if self.per:
                p = priority.cpu().numpy()
                wandb.log({"metrics/mean_priority": np.mean(p), "metrics/max_priority": np.max(p)}, commit=False)
            if self.gpi_pd:
                gp = gpriority.cpu().numpy()
                wandb.log({"metrics/mean_gpriority": np.mean(gp), "metrics/max_gpriority": np.max(gp)}, commit=False)
            wandb.log({"losses/critic_loss": float(th.stack(critic_losses).mean().item()),
                       "metrics/epsilon": self.epsilon, "global_step": self.global_step})

    def last_loss(self) -> float:
        return float(self._out["critic_loss"][0].item())

    # -- acting (gpi_pd.py:564-617) ----------------------------------------------------------------------------------------------
    @th.no_grad()
    def gpi_action(self, obs: th.Tensor, w: th.Tensor, return_policy_index=False, include_w=False):
        sup = self.stacked_weight_support
        if include_w:
            wv = th.as_tensor(w).to(self.engine.q.device, th.float32).reshape(1, -1)
            sup = wv if sup is None else th.cat([sup, wv], dim=0)
        res = self.engine.action(obs, w, sup).cpu()
        if return_policy_index:
            return int(res[0]), int(res[1])
        return int(res[0])

    @th.no_grad()
    def max_action(self, obs: th.Tensor, w: th.Tensor) -> int:
        return int(self.engine.action(obs, w, None)[0].item())

    @th.no_grad()
    def eval(self, obs: np.ndarray, w: np.ndarray) -> int:
        obs = th.as_tensor(obs).float()
        w = th.as_tensor(w).float()
        if self.use_gpi and self.stacked_weight_support is not None:
            return self.gpi_action(obs, w, include_w=False)
        return self.max_action(obs, w)

    @th.no_grad()
    def eval_batch(self, obs: np.ndarray, w: np.ndarray) -> np.ndarray:
        """``eval`` for n (observation, weight) pairs in one pass (lock-step evaluation episodes, ``evaluation.py``)."""
        sup = self.stacked_weight_support if self.use_gpi else None
        return self.engine.actions_rows(th.as_tensor(np.asarray(obs)).float(), th.as_tensor(np.asarray(w)).float(),
                                        sup).cpu().numpy().astype(np.int64)

    def _act(self, obs: th.Tensor, w: th.Tensor) -> int:
        if self.np_random.random() < self.epsilon:
            return self.env.action_space.sample()
        if self.use_gpi and self.stacked_weight_support is not None:
            action, policy_index = self.gpi_action(obs, w, return_policy_index=True)
            self.police_indices.append(policy_index)
            return action
        return self.max_action(obs, w)

    @th.no_grad()
    def _reset_priorities(self, w: th.Tensor):
        """``gpi_pd.py:619-660`` over the device-resident records (chunked so that rows * |M| fits the workspace)."""
        buf, e = self.replay_buffer, self.engine
        buf.flush()
        n = buf.size
        if n == 0:
            return
        M = len(self.weight_support) if self.gpi_pd else 1
        chunk = max(1, min(e.max_rows, (e.max_rows * e.max_support) // max(M, 1)))
        from . import ops
        pri = th.empty(n, dtype=th.float32, device=e.q.device)
        for b in range(0, n, chunk):
            idx = th.arange(b, min(b + chunk, n), dtype=th.int64, device=e.q.device)
            obs, act, rew, nobs, done = ops.gather_batch(self.lib, buf.records, idx, buf._D, buf._R, buf._Ad,
                                                         int_actions=True)
            err = e.priority_errors(obs, act, rew, nobs, done, w, self.stacked_weight_support, gamma=self.gamma,
                                    gpi_pd=self.gpi_pd)
            pri[b:b + idx.numel()] = err.clamp(min=self.min_priority).pow(self.alpha)
        buf.update_priorities(th.arange(n, dtype=th.int64, device=e.q.device), pri)

    def set_weight_support(self, weight_list: List[np.ndarray]):
        weights_no_repeats = unique_tol(weight_list)
        dev = self.engine.q.device
        self.weight_support = [th.tensor(w).float().to(dev) for w in weights_no_repeats]
        self.stacked_weight_support = th.stack(self.weight_support) if self.weight_support else None
        if len(self.weight_support) + 1 > self.engine.max_support * self.engine.max_rows:
            raise ValueError("weight support larger than the engine's workspace (raise max_support)")

    def train_iteration(self, total_timesteps: int, weight: np.ndarray, weight_support: List[np.ndarray],
                        change_w_every_episode: bool = True, reset_num_timesteps: bool = True, eval_env=None,
                        eval_freq: int = 1000, reset_learning_starts: bool = False):
        """``gpi_pd.py:696-789`` (model-free branch)."""
        weight_support = unique_tol(weight_support)
        self.set_weight_support(weight_support)
        dev = self.engine.q.device
        tensor_w = th.tensor(weight).float().to(dev)
        self.police_indices = []
        self.global_step = 0 if reset_num_timesteps else self.global_step
        self.num_episodes = 0 if reset_num_timesteps else self.num_episodes
        if reset_learning_starts:
            self.learning_starts = self.global_step
        if self.per and len(self.replay_buffer) > 0:
            self._reset_priorities(tensor_w)
        obs, info = self.env.reset()
        for _ in range(1, total_timesteps + 1):
            self.global_step += 1
            if self.global_step < self.learning_starts:
                action = self.env.action_space.sample()
            else:
                action = self._act(th.as_tensor(obs).float(), tensor_w)
            next_obs, vec_reward, terminated, truncated, info = self.env.step(action)
            self.replay_buffer.add(obs, action, vec_reward, next_obs, terminated)
            if self.global_step >= self.learning_starts:
                if self.dyna:
                    if self.global_step % self.dynamics_train_freq(self.global_step) == 0:
                        m_obs, m_actions, m_rewards, m_next_obs, m_dones = self.replay_buffer.get_all_data()
                        one_hot = np.zeros((len(m_obs), self.action_dim))
                        one_hot[np.arange(len(m_obs)), m_actions.astype(int).reshape(len(m_obs))] = 1
                        X = np.hstack((m_obs, one_hot))
                        Y = np.hstack((m_rewards, m_next_obs - m_obs))
                        self._last_holdout = self.dynamics.fit(X, Y, **self.dynamics_fit_kwargs)
                    if self.global_step >= self.dynamics_rollout_starts and \
                            self.global_step % self.dynamics_rollout_freq == 0:
                        self._rollout_dynamics(tensor_w)
                self.update(tensor_w)
            if eval_env is not None and self.log and self.global_step % eval_freq == 0:
                self.policy_eval(eval_env, weights=weight, log=self.log)
            if terminated or truncated:
                obs, _ = self.env.reset()
                self.num_episodes += 1
                self.police_indices = []
                if change_w_every_episode:
                    weight = random.choice(weight_support)
                    tensor_w = th.tensor(weight).float().to(dev)
            else:
                obs = next_obs

    def train(self, total_timesteps: int, eval_env, ref_point: np.ndarray, known_pareto_front=None,
              num_eval_weights_for_front: int = 100, num_eval_episodes_for_front: int = 5,
              num_eval_weights_for_eval: int = 50, timesteps_per_iter: int = 10000,
              weight_selection_algo: str = "gpi-ls", eval_freq: int = 1000, eval_mo_freq: int = 10000,
              checkpoints: bool = True):
        """``gpi_pd.py:791-905``: the outer loop is the reference's own ``LinearSupport`` (cvxpy / pycddlib), imported
        unchanged when ``morl_baselines`` is installed -- control plane, not rebuilt here."""
        try:
            from morl_baselines.common.evaluation import log_all_multi_policy_metrics, policy_evaluation_mo
            from morl_baselines.common.weights import equally_spaced_weights
            from morl_baselines.multi_policy.linear_support.linear_support import LinearSupport
        except Exception as exc:  # pragma: no cover
            raise ImportError("train() drives the reference's LinearSupport weight selection: install morl_baselines "
                              "(cvxpy, pycddlib, pymoo); train_iteration() / update() do not need it") from exc
        max_iter = total_timesteps // timesteps_per_iter
        linear_support = LinearSupport(num_objectives=self.reward_dim,
                                       epsilon=0.0 if weight_selection_algo == "ols" else None)
        eval_weights = equally_spaced_weights(self.reward_dim, n=num_eval_weights_for_front)
        for it in range(1, max_iter + 1):
            if weight_selection_algo == "gpi-ls":
                self.set_weight_support(linear_support.get_weight_support())
                use_gpi, self.use_gpi = self.use_gpi, True
                w = linear_support.next_weight(algo="gpi-ls", gpi_agent=self, env=eval_env,
                                               rep_eval=num_eval_episodes_for_front)
                self.use_gpi = use_gpi
            elif weight_selection_algo == "ols":
                w = linear_support.next_weight(algo="ols")
            else:
                raise ValueError(f"Unknown algorithm {weight_selection_algo}.")
            if w is None:
                break
            if weight_selection_algo == "gpi-ls":
                M = linear_support.get_weight_support() + linear_support.get_corner_weights(top_k=4) + [w]
            else:
                M = linear_support.get_weight_support() + [w]
            self.train_iteration(total_timesteps=timesteps_per_iter, weight=w, weight_support=M,
                                 change_w_every_episode=weight_selection_algo == "gpi-ls", eval_env=eval_env,
                                 eval_freq=eval_freq, reset_num_timesteps=False, reset_learning_starts=False)
            if weight_selection_algo == "ols":
                linear_support.add_solution(policy_evaluation_mo(self, eval_env, w, rep=num_eval_episodes_for_front)[3], w)
            else:
                for wcw in M:
                    linear_support.add_solution(
                        policy_evaluation_mo(self, eval_env, wcw, rep=num_eval_episodes_for_front)[3], wcw)
            if self.log and self.global_step % eval_mo_freq == 0:
                front = front_returns(self, eval_env, eval_weights, rep=num_eval_episodes_for_front)
                log_all_multi_policy_metrics(current_front=front, hv_ref_point=ref_point, reward_dim=self.reward_dim,
                                             global_step=self.global_step, n_sample_weights=num_eval_weights_for_eval,
                                             ref_front=known_pareto_front)
            if checkpoints:
                self.save(filename=f"GPI-PD {weight_selection_algo} iter={it}", save_replay_buffer=False)
        self.close_wandb()


class GPILS(GPIPD):
    """Model-free GPI-LS (``gpi_pd.py:908-916``): no dynamics model, plain TD priorities."""

    def __init__(self, *args, **kwargs):
        if "experiment_name" not in kwargs:
            kwargs["experiment_name"] = "GPI-LS"
        kwargs.pop("dyna", None)
        kwargs.pop("gpi_pd", None)
        super().__init__(*args, dyna=False, gpi_pd=False, **kwargs)
