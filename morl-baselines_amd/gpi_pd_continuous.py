"""GPI-PD / GPI-LS with continuous actions (TD3 style) on the HIP actor-critic engine
(``multi_policy/gpi_pd/gpi_pd_continuous_action.py``).

``update(weight)`` runs each of the ``gradient_updates`` iterations as one ``morl_ac_update``: target-policy smoothing,
twin LayerNorm + Dropout target critics, ``argmin_n (Q_n . w)`` target selection, critic loss / backward / Adam, PER
priorities, Polyak, and every ``delay_policy_update``-th iteration the deterministic actor step through the critics
plus the target-policy Polyak.  The GPI evaluation (``eval`` with ``use_gpi``: |M| policy actions x |M| weights through
the first critic, ``max_a`` then ``argmax_i``) runs the |M|^2 critic rows through ``morl_ac_q_forward``.

``dyna=True`` (GPI-PD proper): the probabilistic dynamics ensemble is trained on the device (``dynamics.py``,
``morl_ens_*``), roll-outs from replayed states act with the noisy TD3 policy (one ``morl_ac_policy_forward`` per plan
step), the uncertainty-filtered imagined transitions go into the model buffer with one batched add, and updates draw
real / imagined mixes (``gpi_pd_continuous_action.py:313-371, 548-561``).  ``GPILSContinuousAction`` is the model-free
flavour.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Union

import numpy as np
import torch as th

from .evaluation import front_returns
from .ac_engine import ALGO_TD3, ACEngine
from .acnets import PolicyShell, QNetworkShell, adam_state_dict, as_f32, bind, load_adam_state_dict, randn
from .api import MOAgent, MOPolicy
from .native import NativeLib, load_library
from .replay import PrioritizedReplayBuffer, ReplayBuffer


def unique_tol(a: List[np.ndarray], tol=1e-4) -> List[np.ndarray]:
    """``common/utils.py`` ``unique_tol``: drop vectors that repeat an earlier one up to ``tol``."""
    if len(a) == 0:
        return a
    delete = np.array([False] * len(a))
    a = np.array(a)
    for i in range(len(a)):
        if delete[i]:
            continue
        for j in range(i + 1, len(a)):
            if np.allclose(a[i], a[j], tol):
                delete[j] = True
    return list(a[~delete])


class GPIPDContinuousAction(MOAgent, MOPolicy):
    """GPI-PD with continuous actions (Alegre et al., AAMAS 2023, appendix) -- model-free path on the MI355X."""

    def __init__(self, env, learning_rate: float = 3e-4, gamma: float = 0.99, tau: float = 0.005,
                 buffer_size: int = 400000, net_arch: List = [256, 256], batch_size: int = 128, num_q_nets: int = 2,
                 delay_policy_update: int = 2, learning_starts: int = 100, gradient_updates: int = 20,
                 use_gpi: bool = False, policy_noise: float = 0.2, noise_clip: float = 0.5, per: bool = True,
                 min_priority: float = 0.1, alpha: float = 0.6, dyna: bool = True,
                 dynamics_net_arch: List = [200, 200, 200, 200], dynamics_train_freq: int = 250,
                 dynamics_rollout_len: int = 5, dynamics_rollout_starts: int = 1000, dynamics_rollout_freq: int = 250,
                 dynamics_rollout_batch_size: int = 50000, dynamics_buffer_size: int = 200000,
                 dynamics_min_uncertainty: float = 2.0, dynamics_real_ratio: float = 0.1,
                 project_name: str = "MORL-Baselines", experiment_name: str = "GPI-PD Continuous Action",
                 wandb_entity: Optional[str] = None, log: bool = True, seed: Optional[int] = None,
                 device: Union[th.device, str] = "auto", lib: Optional[NativeLib] = None,
                 q_layer_norm: bool = True, q_drop_rate: float = 0.01, dynamics_max_rows: int = 10000,
                 model_termination_func=None):
        MOAgent.__init__(self, env, device=device, seed=seed)
        MOPolicy.__init__(self, device=device)
        self.learning_rate, self.tau, self.gamma, self.use_gpi = learning_rate, tau, gamma, use_gpi
        self.policy_noise, self.noise_clip, self.buffer_size = policy_noise, noise_clip, buffer_size
        self.num_q_nets, self.delay_policy_update, self.net_arch = num_q_nets, delay_policy_update, net_arch
        self.learning_starts, self.batch_size, self.gradient_updates = learning_starts, batch_size, gradient_updates
        self.per, self.min_priority, self.alpha, self.dyna = per, min_priority, alpha, dyna
        self.lib = lib or load_library()
        buf_cls = PrioritizedReplayBuffer if per else ReplayBuffer
        self.replay_buffer = buf_cls(self.observation_shape, self.action_dim, rew_dim=self.reward_dim,
                                     max_size=buffer_size, device=self.device, lib=self.lib)
        low, high = np.asarray(env.action_space.low), np.asarray(env.action_space.high)
        self.engine = ACEngine(ALGO_TD3, self.observation_dim, self.action_dim, self.reward_dim, net_arch,
                               action_low=low, action_high=high, max_rows=2 * batch_size, num_q=num_q_nets,
                               q_layer_norm=q_layer_norm, q_drop_rate=q_drop_rate, device=self.device, lib=self.lib)
        e = self.engine
        qin = self.observation_dim + self.action_dim + self.reward_dim
        mk_q = lambda: QNetworkShell(qin, self.reward_dim, net_arch, drop_rate=q_drop_rate, layer_norm=q_layer_norm)  # noqa: E731
        self.q_nets = [mk_q() for _ in range(num_q_nets)]
        self.target_q_nets = [mk_q() for _ in range(num_q_nets)]
        pin = self.observation_dim + self.reward_dim
        self.policy = PolicyShell(pin, self.action_dim, net_arch, ("mean",), low, high)
        self.target_policy = PolicyShell(pin, self.action_dim, net_arch, ("mean",), low, high)
        for n in range(num_q_nets):
            bind(self.q_nets[n], e.q_views(e.q, 0, n))
            bind(self.target_q_nets[n], e.q_views(e.q_target, 0, n), copy_in=False)
        bind(self.policy, e.policy_views(e.pol))
        bind(self.target_policy, e.policy_views(e.pol_target), copy_in=False)
        e.q_target.copy_(e.q)
        e.pol_target.copy_(e.pol)
        # the Dyna model (gpi_pd_continuous_action.py:216-235): input (obs, action), output (reward, next_obs - obs)
        self.dynamics_net_arch, self.dynamics, self.dynamics_buffer = dynamics_net_arch, None, None
        if self.dyna:
            from .dynamics import ProbabilisticEnsemble
            self.dynamics = ProbabilisticEnsemble(input_dim=self.observation_dim + self.action_dim,
                                                  output_dim=self.observation_dim + self.reward_dim,
                                                  arch=self.dynamics_net_arch, device=self.device, lib=self.lib,
                                                  max_rows=dynamics_max_rows)   # >= fit batch / hold-out / roll-out rows
            self.dynamics_buffer = ReplayBuffer(self.observation_shape, self.action_dim, rew_dim=self.reward_dim,
                                                max_size=dynamics_buffer_size, device=self.device, lib=self.lib)
        self.dynamics_train_freq, self.dynamics_rollout_len = dynamics_train_freq, dynamics_rollout_len
        self.dynamics_rollout_starts, self.dynamics_rollout_freq = dynamics_rollout_starts, dynamics_rollout_freq
        self.dynamics_rollout_batch_size, self.dynamics_min_uncertainty = dynamics_rollout_batch_size, dynamics_min_uncertainty
        self.dynamics_real_ratio, self.dynamics_max_rows = dynamics_real_ratio, dynamics_max_rows
        self.dynamics_fit_kwargs = {}                 # forwarded to ProbabilisticEnsemble.fit (reference: defaults)
        self.model_termination_func = model_termination_func   # None: looked up from the environment id like ModelEnv does
        self._last_rollout, self._last_holdout = None, None
        self.weight_support: List[th.Tensor] = []
        self.stacked_weight_support = []
        self._n_updates = 0
        self._q_step = self._p_step = 0
        self._drop_seed = (0 if seed is None else int(seed)) * 1000003 + 12345   # never touches self.np_random
        self._out = None
        self.experiment_name = experiment_name
        self.log = log
        if self.log:
            self.setup_wandb(project_name, experiment_name, wandb_entity)

    def get_config(self):
        return {"env_id": self.env.unwrapped.spec.id, "learning_rate": self.learning_rate,
                "num_q_nets": self.num_q_nets, "batch_size": self.batch_size, "tau": self.tau, "gamma": self.gamma,
                "net_arch": self.net_arch, "use_gpi": self.use_gpi, "policy_noise": self.policy_noise,
                "noise_clip": self.noise_clip, "gradient_updates": self.gradient_updates,
                "delay_policy_update": self.delay_policy_update, "min_priority": self.min_priority, "per": self.per,
                "buffer_size": self.buffer_size, "alpha": self.alpha, "learning_starts": self.learning_starts,
                "dyna": self.dyna, "seed": self.seed}

    # -- checkpoints (gpi_pd_continuous_action.py:276-312) ---------------------------------------------------------------
    def _q_views(self, buf):
        return [v for n in range(self.num_q_nets) for v in self.engine.q_views(buf, 0, n)]

    def save(self, save_dir="weights/", filename=None, save_replay_buffer=True):
        if not os.path.isdir(save_dir):
            os.makedirs(save_dir)
        e = self.engine
        saved = {"policy_state_dict": self.policy.state_dict(),
                 "policy_optimizer_state_dict": adam_state_dict(e.policy_views(e.pol), e.policy_views(e.pol_exp_avg),
                                                                e.policy_views(e.pol_exp_avg_sq), self._p_step,
                                                                self.learning_rate)}
        for i, (q, tq) in enumerate(zip(self.q_nets, self.target_q_nets)):
            saved["q_net_" + str(i) + "_state_dict"] = q.state_dict()
            saved["target_q_net_" + str(i) + "_state_dict"] = tq.state_dict()
        saved["q_nets_optimizer_state_dict"] = adam_state_dict(self._q_views(e.q), self._q_views(e.q_exp_avg),
                                                               self._q_views(e.q_exp_avg_sq), self._q_step,
                                                               self.learning_rate)
        saved["M"] = self.weight_support
        if self.dyna:
            saved["dynamics_state_dict"] = self.dynamics.state_dict()
        saved["target_policy_state_dict"] = self.target_policy.state_dict()   # (extension: not in the reference file)
        if save_replay_buffer:
            saved["replay_buffer"] = self.replay_buffer
        filename = self.experiment_name if filename is None else filename
        th.save(saved, save_dir + "/" + filename + ".tar")

    def load(self, path, load_replay_buffer=True):
        params = th.load(path, map_location=self.device, weights_only=False)
        e = self.engine
        self.weight_support = [w.to(e.q.device) for w in params["M"]]
        if self.weight_support:
            self.stacked_weight_support = th.stack(self.weight_support)
        self.policy.load_state_dict(params["policy_state_dict"])
        if "target_policy_state_dict" in params:
            self.target_policy.load_state_dict(params["target_policy_state_dict"])
        self._p_step = load_adam_state_dict(params["policy_optimizer_state_dict"], e.policy_views(e.pol_exp_avg),
                                            e.policy_views(e.pol_exp_avg_sq))
        for i, (q, tq) in enumerate(zip(self.q_nets, self.target_q_nets)):
            q.load_state_dict(params["q_net_" + str(i) + "_state_dict"])
            tq.load_state_dict(params["target_q_net_" + str(i) + "_state_dict"])
        self._q_step = load_adam_state_dict(params["q_nets_optimizer_state_dict"], self._q_views(e.q_exp_avg),
                                            self._q_views(e.q_exp_avg_sq))
        if self.dyna and "dynamics_state_dict" in params:
            self.dynamics.load_state_dict(params["dynamics_state_dict"])
        if load_replay_buffer and "replay_buffer" in params:
            self.replay_buffer = params["replay_buffer"]

    def _sample_batch_experiences(self):
        """``gpi_pd_continuous_action.py:313-335``: real transitions, or a real / imagined mix once the model buffer is used."""
        if not self.dyna or self.global_step < self.dynamics_rollout_starts or len(self.dynamics_buffer) == 0:
            return self.replay_buffer.sample(self.batch_size, to_tensor=True, device=self.device)
        num_real = int(self.batch_size * self.dynamics_real_ratio)
        real = self.replay_buffer.sample(num_real, to_tensor=True, device=self.device)
        model = self.dynamics_buffer.sample(self.batch_size - num_real, to_tensor=True, device=self.device)
        mixed = tuple(th.cat([r.reshape(r.shape[0], -1), m.reshape(m.shape[0], -1)], dim=0)
                      for r, m in zip(real[:5], model[:5]))
        return mixed + ((real[5],) if self.per else ())

    @th.no_grad()
    def _rollout_dynamics(self, weight: th.Tensor):
        """``gpi_pd_continuous_action.py:336-371``: imagined transitions from the noisy policy under the current weight."""
        from .dynamics import ModelEnv
        e = self.engine
        dev = e.q.device
        num_times = int(np.ceil(self.dynamics_rollout_batch_size / 10000))
        batch_size = min(self.dynamics_rollout_batch_size, 10000)
        weight = as_f32(weight, dev).reshape(1, -1)
        cfg = e.make_cfg(policy_noise=self.policy_noise, noise_clip=self.noise_clip)
        added, unc = 0, np.zeros(1)
        for _ in range(num_times):
            obs = self.replay_buffer.sample_obs(batch_size, to_tensor=False)
            model_env = ModelEnv(self.dynamics, self.env.unwrapped.spec.id, rew_dim=self.reward_dim,
                                 termination_func=self.model_termination_func)
            for _plan_step in range(self.dynamics_rollout_len):
                obs_t = as_f32(np.ascontiguousarray(obs, dtype=np.float32), dev)
                n = obs_t.shape[0]
                w = weight.repeat(n, 1)
                # Policy.forward with noise (:50-58): one th.randn_like draw per call, the whole batch at once
                noise = randn((n, self.action_dim), dev)
                actions = th.cat([e.policy_forward(obs_t[b:b + e.max_rows].contiguous(), w[b:b + e.max_rows].contiguous(),
                                                   eps=noise[b:b + e.max_rows].contiguous(), cfg=cfg)[0]
                                  for b in range(0, n, e.max_rows)], dim=0)
                next_obs_pred, r_pred, dones, info = model_env.step(obs_t, actions)
                unc = info["uncertainty"]
                keep = unc < self.dynamics_min_uncertainty
                obs_h, act_h = obs_t.cpu().numpy(), actions.cpu().numpy()
                self.dynamics_buffer.add_batch(obs_h[keep], act_h[keep], r_pred[keep], next_obs_pred[keep], dones[keep])
                added += int(keep.sum())
                nonterm = ~dones.squeeze(-1)
                if nonterm.sum() == 0:
                    break
                obs = next_obs_pred[nonterm]
        self._last_rollout = {"imagined": added, "uncertainty_mean": float(unc.mean()), "uncertainty_max": float(unc.max()),
                              "uncertainty_min": float(unc.min())}
        if self.log:
            import wandb
            wandb.log({"dynamics/uncertainty_mean": unc.mean(), "dynamics/uncertainty_max": unc.max(),
                       "dynamics/uncertainty_min": unc.min(), "global_step": self.global_step})

    # -- the hot path (gpi_pd_continuous_action.py:373-452) ---------------------------------------------------------------
    per_one_entry_enabled = True      # False: one sample / update / update_priorities entry per iteration (the A/B of the tests)

    def update(self, weight: th.Tensor):
        e = self.engine
        dev = e.q.device
        weight = as_f32(weight, dev).reshape(-1)
        priority, deferred = None, []
        real_only = not self.dyna or self.global_step < self.dynamics_rollout_starts or len(self.dynamics_buffer) == 0
        # (one tree-update launch holds ST_MAX_B = 1 024 entries: larger batches keep the per-iteration rounds, whose
        # update_priorities splits the update into blocks)
        per_one_entry = self.per_one_entry_enabled and self.per and real_only and not self.replay_buffer._int_actions and \
            self.batch_size <= self.replay_buffer.TREE_BLOCK
        B = self.batch_size
        doubled = len(self.weight_support) > 1
        if per_one_entry:
            # the whole loop as ONE library entry (morl_ac_update_n_per): the host draws what the reference's loop draws, in its
            # order per generator (numpy: the B unit uniforms of each PrioritizedReplayBuffer.sample; random: choices; torch: the
            # target noise) -- none of it depends on what the device computes
            rows = 2 * B if doubled else B
            sc = self.__dict__.get("_per_scratch")
            if sc is None or sc[0].shape[0] != rows:
                D, R, Ad = self.replay_buffer._D, self.replay_buffer._R, self.replay_buffer._Ad
                sc = (th.empty((rows, D), dtype=th.float32, device=dev), th.empty((rows, Ad), dtype=th.float32, device=dev),
                      th.empty((rows, R), dtype=th.float32, device=dev), th.empty((rows, D), dtype=th.float32, device=dev),
                      th.empty((rows, 1), dtype=th.float32, device=dev))
                self._per_scratch = sc
            u01 = np.empty((self.gradient_updates, B), dtype=np.float64)
        for g in range(self.gradient_updates):
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
                    s_obs, s_actions, s_rewards, s_next_obs, s_dones = (x.repeat(2, 1) for x in
                                                                        (s_obs, s_actions, s_rewards, s_next_obs, s_dones))
                w = th.vstack([weight.expand(B, -1)] + random.choices(self.weight_support, k=B))
            else:
                w = weight.repeat(B, 1)
            rows = s_obs.size(0)
            do_policy = self._n_updates % self.delay_policy_update == 0
            self._q_step += 1
            if do_policy:
                self._p_step += 1
            self._drop_seed += 1
            cfg = e.make_cfg(gamma=self.gamma, tau=self.tau, q_lr=self.learning_rate, policy_lr=self.learning_rate,
                             q_step=self._q_step, policy_step=self._p_step, do_policy=do_policy,
                             policy_noise=self.policy_noise, noise_clip=self.noise_clip, n_per=(n_per if self.per else 0),
                             dropout_seed=self._drop_seed)
            noise = randn((rows, self.action_dim), dev)
            want = ("critic_loss",) + (("policy_loss",) if do_policy else ()) + (("priority",) if self.per else ())
            kw = dict(obs=s_obs, actions=s_actions, rewards=s_rewards, next_obs=s_next_obs, dones=s_dones.reshape(-1), w=w,
                      eps_next=noise, want=want)
            self._n_updates += 1
            if not self.per or per_one_entry:
                # no prioritised replay: nothing of iteration g feeds the sampling of iteration g + 1 -- the loop is drawn first
                # and submitted as ONE library entry below (morl_ac_update_n)
                deferred.append(dict(kw, cfg=cfg))
                continue
            out = e.update(cfg, **kw)
            self._out = {**(self._out or {}), **out}
            priority = out["priority"][0].clamp(min=self.min_priority).pow(self.alpha)
            self.replay_buffer.update_priorities(idxes, priority)
        if deferred:
            if per_one_entry:
                outs, self._last_per_idx = e.update_n_per(deferred, buffer=self.replay_buffer, u01=u01, doubled=doubled,
                                                          alpha=self.alpha, min_priority=self.min_priority)
                priority = outs[-1]["priority"][0].clamp(min=self.min_priority).pow(self.alpha)
            elif len(deferred) == 1:
                kw = deferred[0]
                outs = [e.update(kw.pop("cfg"), **kw)]
            else:
                outs = e.update_n(deferred)
            for out in outs:
                self._out = {**(self._out or {}), **out}
        if self.log and self.global_step % 100 == 0:
            import wandb
            if self.per:
                p = priority.cpu().numpy()
                wandb.log({"metrics/mean_priority": np.mean(p), "metrics/max_priority": np.max(p),
                           "metrics/min_priority": np.min(p)}, commit=False)
            wandb.log({"losses/critic_loss": float(self._out["critic_loss"][0].item()),
                       "losses/policy_loss": float(self._out["policy_loss"][0].item()),
                       "global_step": self.global_step})

    @th.no_grad()
    def eval(self, obs: Union[np.ndarray, th.Tensor], w: Union[np.ndarray, th.Tensor], torch_action=False):
        """``gpi_pd_continuous_action.py:454-484``."""
        e = self.engine
        dev = e.q.device
        obs, w = as_f32(obs, dev).reshape(-1), as_f32(w, dev).reshape(-1)
        if self.use_gpi and len(self.weight_support) > 0:
            M = self.stacked_weight_support
            m = M.size(0)
            cap = e.max_rows
            actions_original = th.cat([e.policy_forward(obs.expand(min(cap, m - b), -1).contiguous(), M[b:b + cap].contiguous())[0]
                                       for b in range(0, m, cap)], dim=0)                           # (m, Ad): pi(s, w_i)
            # critic 0 at every (policy action a_i, conditioning weight w_p) pair: row p * m + i -- |M|^2 rows (4096 at the
            # 64-weight GPI set of BASELINE config 3), streamed through the engine's workspace in chunks of max_rows
            acts_rows, w_rows = actions_original.repeat(m, 1), M.repeat_interleave(m, dim=0)
            q = th.cat([e.q_forward(obs.expand(min(cap, m * m - b), -1).contiguous(), acts_rows[b:b + cap].contiguous(),
                                    w_rows[b:b + cap].contiguous())[0, 0] for b in range(0, m * m, cap)], dim=0)
            q = q.view(m, m, self.reward_dim)
            scalar_values = th.einsum("par,r->pa", q, w)
            max_q, a = th.max(scalar_values, dim=1)
            action = actions_original[a[th.argmax(max_q)]]
        else:
            action = e.policy_forward(obs.reshape(1, -1), w.reshape(1, -1))[0, 0]
        return action if torch_action else action.detach().cpu().numpy()

    def set_weight_support(self, weight_list: List[np.ndarray]):
        weights_no_repeat = unique_tol(weight_list)
        dev = self.engine.q.device
        self.weight_support = [th.tensor(w).float().to(dev) for w in weights_no_repeat]
        if len(self.weight_support) > 0:
            self.stacked_weight_support = th.stack(self.weight_support)

    @th.no_grad()
    def _explore_action(self, obs, tensor_w) -> np.ndarray:
        e = self.engine
        o = as_f32(np.asarray(obs, dtype=np.float32), e.q.device).reshape(1, -1)
        noise = randn((1, self.action_dim), e.q.device)
        cfg = e.make_cfg(policy_noise=self.policy_noise, noise_clip=self.noise_clip)
        return e.policy_forward(o, tensor_w.reshape(1, -1), eps=noise, cfg=cfg)[0, 0].cpu().numpy()

    def train_iteration(self, total_timesteps: int, weight: np.ndarray, weight_support: List[np.ndarray],
                        change_weight_every_episode: bool = False, eval_env=None, eval_freq: int = 1000,
                        reset_num_timesteps: bool = False):
        """``gpi_pd_continuous_action.py:493-584``."""
        weight_support = unique_tol(weight_support)
        self.set_weight_support(weight_support)
        dev = self.engine.q.device
        tensor_w = th.tensor(weight).float().to(dev)
        self.global_step = 0 if reset_num_timesteps else self.global_step
        self.num_episodes = 0 if reset_num_timesteps else self.num_episodes
        obs, info = self.env.reset()
        for _ in range(1, total_timesteps + 1):
            self.global_step += 1
            if self.global_step < self.learning_starts:
                action = self.env.action_space.sample()
            else:
                action = self._explore_action(obs, tensor_w)
            next_obs, vector_reward, terminated, truncated, info = self.env.step(action)
            self.replay_buffer.add(obs, action, vector_reward, next_obs, terminated)
            if self.global_step >= self.learning_starts:
                if self.dyna:
                    if self.global_step % self.dynamics_train_freq == 0:
                        m_obs, m_actions, m_rewards, m_next_obs, _ = self.replay_buffer.get_all_data()
                        X = np.hstack((m_obs, m_actions))
                        Y = np.hstack((m_rewards, m_next_obs - m_obs))
                        self._last_holdout = self.dynamics.fit(X, Y, **self.dynamics_fit_kwargs)
                        if self.log:
                            import wandb
                            wandb.log({"dynamics/mean_holdout_loss": self._last_holdout, "global_step": self.global_step})
                    if self.global_step >= self.dynamics_rollout_starts and \
                            self.global_step % self.dynamics_rollout_freq == 0:
                        self._rollout_dynamics(tensor_w)
                self.update(tensor_w)
            if eval_env is not None and self.log and self.global_step % eval_freq == 0:
                self.policy_eval(eval_env, weights=weight, log=self.log)
            if terminated or truncated:
                obs, _ = self.env.reset()
                self.num_episodes += 1
                if change_weight_every_episode:
                    weight = random.choice(weight_support)
                    tensor_w = th.tensor(weight).float().to(dev)
            else:
                obs = next_obs

    def train(self, total_timesteps: int, eval_env, ref_point: np.ndarray, known_pareto_front=None,
              num_eval_weights_for_front: int = 100, num_eval_episodes_for_front: int = 5,
              num_eval_weights_for_eval: int = 50, weight_selection_algo: str = "gpi-ls",
              timesteps_per_iter: int = 10000, eval_freq: int = 1000, eval_mo_freq: int = 10000,
              checkpoints: bool = True):
        """``gpi_pd_continuous_action.py:586-708``: the outer weight-selection loop is the reference's own
        ``LinearSupport`` (cvxpy / pycddlib) -- control plane, imported unchanged when available."""
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
                                 change_weight_every_episode=weight_selection_algo == "gpi-ls", eval_env=eval_env,
                                 eval_freq=eval_freq)
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


class GPILSContinuousAction(GPIPDContinuousAction):
    """Model-free GPI-LS with continuous actions (``gpi_pd_continuous_action.py:711-718``)."""

    def __init__(self, *args, **kwargs):
        if "experiment_name" not in kwargs:
            kwargs["experiment_name"] = "GPI-LS Continuous Action"
        kwargs.pop("dyna", None)
        super().__init__(*args, dyna=False, **kwargs)
