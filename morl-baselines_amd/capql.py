"""CAPQL on the HIP actor-critic engine, with the reference's class surface (``multi_policy/capql/capql.py``).

Same constructor arguments, ``update()`` / ``eval()`` / ``train()`` / ``save()`` / ``load()`` / ``get_config()``,
``ReplayMemory`` and ``WeightSamplerAngle``.  What moved to the device:

* ``ReplayMemory`` mirrors every pushed transition (state | action | weights | reward | next_state | done) into one
  device record; ``sample`` draws the indices on the host exactly as the reference does (``random.sample`` over the
  list -> same ``random`` stream) and gathers on the device (``morl_gather_fields``);
* ``update()`` is one ``morl_ac_update`` call per gradient step: policy sample at s', twin target critics, TD target,
  twin critics forward/backward, Adam, actor loss through the updated critics, Adam, Polyak -- no host
  synchronisation (the losses stay on the device until somebody logs them).
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Union

import numpy as np
import torch as th

from . import ops
from .evaluation import front_returns
from .ac_engine import ALGO_CAPQL, ACEngine
from .acnets import PolicyShell, QNetworkShell, adam_state_dict, as_f32, bind, load_adam_state_dict, randn
from .api import MOAgent, MOPolicy
from .native import NativeLib, load_library


class ReplayMemory:
    """``capql.py:32-66`` with a device mirror.  ``buffer`` stays the reference's list of 6-tuples."""

    _PENDING = 1024

    def __init__(self, capacity: int, device="cuda", lib: Optional[NativeLib] = None):
        self.capacity = capacity
        self.buffer = []
        self.position = 0
        self.device = th.device(device)
        self.lib = lib or load_library()
        self.records = None
        self._fields = None

    def _init_device(self, sizes):
        self._sizes = sizes                                   # state, action, weights, reward, next_state, done
        offs = np.concatenate([[0], np.cumsum(sizes)])
        self._fields = [(int(offs[k]), int(sizes[k])) for k in range(6)]
        self._rec = int(offs[-1])
        self.records = th.zeros((self.capacity, self._rec), dtype=th.float32, device=self.device)
        self._stage = th.zeros((self._PENDING, self._rec), dtype=th.float32, pin_memory=self.device.type == "cuda")
        self._stage_np = self._stage.numpy()
        self._pending_start, self._pending_n = 0, 0

    def push(self, state, action, weights, reward, next_state, done):
        item = (np.array(state).copy(), np.array(action).copy(), np.array(weights).copy(), np.array(reward).copy(),
                np.array(next_state).copy(), np.array(done).copy())
        if self.records is None:
            self._init_device([int(np.size(x)) for x in item])
        if len(self.buffer) < self.capacity:
            self.buffer.append(None)
        self.buffer[self.position] = item
        if self._pending_n == 0:
            self._pending_start = self.position
        elif self._pending_n == self._PENDING or (self._pending_start + self._pending_n) % self.capacity != self.position:
            self.flush()
            self._pending_start = self.position
        row = self._stage_np[self._pending_n]
        for (o, w_), x in zip(self._fields, item):
            row[o:o + w_] = np.asarray(x, dtype=np.float32).reshape(-1)
        self._pending_n += 1
        self.position = (self.position + 1) % self.capacity

    def flush(self):
        n = self._pending_n
        if n == 0:
            return
        start = self._pending_start
        first = min(n, self.capacity - start)
        # blocking copies: the staging rows are rewritten by the very next push
        self.records[start:start + first].copy_(self._stage[:first])
        if first < n:
            self.records[:n - first].copy_(self._stage[first:n])
        self._pending_n = 0

    def sample(self, batch_size, to_tensor=True, device=None):
        """Same ``random`` stream as ``random.sample(self.buffer, batch_size)`` (the selection depends only on the
        population size), device gather when ``to_tensor``."""
        inds = random.sample(range(len(self.buffer)), batch_size)
        if not to_tensor:
            batch = [self.buffer[i] for i in inds]
            return tuple(map(np.stack, zip(*batch)))
        self.flush()
        idx = th.as_tensor(inds, dtype=th.int64).to(self.device, non_blocking=True)
        state, action, w, reward, next_state, done = ops.gather_fields(self.lib, self.records, idx, self._fields)
        return state, action, w, reward, next_state, done.reshape(-1)

    def __len__(self):
        return len(self.buffer)

    def __getstate__(self):
        return {"capacity": self.capacity, "buffer": self.buffer, "position": self.position, "device": str(self.device)}

    def __setstate__(self, st):
        self.__init__(st["capacity"], device=st["device"] if th.cuda.is_available() else "cpu")
        for item in st["buffer"]:
            self.push(*item)
        self.position = st["position"]
        self.flush()


class WeightSamplerAngle:
    """``capql.py:69-97``: weight vectors within ``angle`` of ``w`` (torch global RNG, host side)."""

    def __init__(self, rwd_dim, angle, w=None):
        self.rwd_dim = rwd_dim
        self.angle = angle
        if w is None:
            w = th.ones(rwd_dim)
        self.w = w / th.norm(w)

    def sample(self, n_sample):
        s = th.normal(th.zeros(n_sample, self.rwd_dim))
        s = s - (s @ self.w).view(-1, 1) * self.w.view(1, -1)        # drop the component along w
        s = s / th.norm(s, dim=1, keepdim=True)
        s_angle = th.rand(n_sample, 1) * self.angle
        w_sample = th.tan(s_angle) * s + self.w.view(1, -1)
        w_sample = w_sample / th.norm(w_sample, dim=1, keepdim=True, p=1)
        return w_sample.float()


class CAPQL(MOAgent, MOPolicy):
    """CAPQL (Lu, Herman & Yu, ICLR 2023) -- ``capql.py:176-517`` on the MI355X engine."""

    def __init__(self, env, learning_rate: float = 3e-4, gamma: float = 0.99, tau: float = 0.005,
                 buffer_size: int = 1000000, net_arch: List = [256, 256], batch_size: int = 128, num_q_nets: int = 2,
                 alpha: float = 0.2, learning_starts: int = 1000, gradient_updates: int = 1,
                 project_name: str = "MORL-Baselines", experiment_name: str = "CAPQL",
                 wandb_entity: Optional[str] = None, log: bool = True, seed: Optional[int] = None,
                 device: Union[th.device, str] = "auto", lib: Optional[NativeLib] = None):
        MOAgent.__init__(self, env, device=device, seed=seed)
        MOPolicy.__init__(self, device=device)
        self.learning_rate, self.tau, self.gamma = learning_rate, tau, gamma
        self.buffer_size, self.num_q_nets, self.net_arch = buffer_size, num_q_nets, net_arch
        self.learning_starts, self.batch_size, self.gradient_updates = learning_starts, batch_size, gradient_updates
        self.alpha = alpha
        self.lib = lib or load_library()
        self.replay_buffer = ReplayMemory(self.buffer_size, device=self.device, lib=self.lib)
        low, high = np.asarray(self.env.action_space.low), np.asarray(self.env.action_space.high)
        self.engine = ACEngine(ALGO_CAPQL, self.observation_dim, self.action_dim, self.reward_dim, net_arch,
                               action_low=low, action_high=high, max_rows=batch_size, num_q=num_q_nets,
                               device=self.device, lib=self.lib)
        e = self.engine
        qin = self.observation_dim + self.action_dim + self.reward_dim
        # construction order (and therefore torch-RNG consumption) of capql.py:247-264
        self.q_nets = [QNetworkShell(qin, self.reward_dim, net_arch) for _ in range(num_q_nets)]
        self.target_q_nets = [QNetworkShell(qin, self.reward_dim, net_arch) for _ in range(num_q_nets)]
        self.policy = PolicyShell(self.observation_dim + self.reward_dim, self.action_dim, net_arch,
                                  ("mean", "log_std_linear"), low, high)
        for n in range(num_q_nets):
            bind(self.q_nets[n], e.q_views(e.q, 0, n))
            bind(self.target_q_nets[n], e.q_views(e.q_target, 0, n), copy_in=False)
        e.q_target.copy_(e.q)                                    # load_state_dict(q_net.state_dict())
        bind(self.policy, e.policy_views(e.pol))
        self._q_step = self._p_step = 0
        self._n_updates = 0
        self._out = None
        self.log = log
        if self.log:
            self.setup_wandb(project_name, experiment_name, wandb_entity)

    def get_config(self):
        return {"env_id": self.env.unwrapped.spec.id, "learning_rate": self.learning_rate,
                "num_q_nets": self.num_q_nets, "batch_size": self.batch_size, "tau": self.tau, "gamma": self.gamma,
                "net_arch": self.net_arch, "gradient_updates": self.gradient_updates, "alpha": self.alpha,
                "buffer_size": self.buffer_size, "learning_starts": self.learning_starts, "seed": self.seed}

    # -- checkpoints (capql.py:295-325; optimiser states in torch.optim.Adam.state_dict() layout) ------------------------
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
        if save_replay_buffer:
            saved["replay_buffer"] = self.replay_buffer
        filename = self.experiment_name if filename is None else filename
        th.save(saved, save_dir + "/" + filename + ".tar")

    def load(self, path, load_replay_buffer=True):
        params = th.load(path, map_location=self.device, weights_only=False)
        e = self.engine
        self.policy.load_state_dict(params["policy_state_dict"])
        self._p_step = load_adam_state_dict(params["policy_optimizer_state_dict"], e.policy_views(e.pol_exp_avg),
                                            e.policy_views(e.pol_exp_avg_sq))
        for i, (q, tq) in enumerate(zip(self.q_nets, self.target_q_nets)):
            q.load_state_dict(params["q_net_" + str(i) + "_state_dict"])
            tq.load_state_dict(params["target_q_net_" + str(i) + "_state_dict"])
        self._q_step = load_adam_state_dict(params["q_nets_optimizer_state_dict"], self._q_views(e.q_exp_avg),
                                            self._q_views(e.q_exp_avg_sq))
        if load_replay_buffer and "replay_buffer" in params:
            self.replay_buffer = params["replay_buffer"]

    def _sample_batch_experiences(self):
        return self.replay_buffer.sample(self.batch_size, to_tensor=True, device=self.device)

    # -- the hot path (capql.py:321-362) ----------------------------------------------------------------------------------
    def update(self):
        """``capql.py:321-362``.  The batches and the noise of all ``gradient_updates`` iterations are drawn first (each RNG
        stream -- Python's ``random`` for the batch, torch's generator for the two ``rsample()`` draws -- is consumed in the
        reference's order), then the whole loop is ONE library entry (``morl_ac_update_n``)."""
        e = self.engine
        items = []
        for _ in range(self.gradient_updates):
            s_obs, s_actions, w, s_rewards, s_next_obs, s_dones = self._sample_batch_experiences()
            B = s_obs.shape[0]
            # two draws shaped like the reference's two rsample() calls (capql.py:326, :341): on the CPU test backend
            # this consumes torch's generator exactly as the reference does
            eps = (randn((B, self.action_dim), e.q.device), randn((B, self.action_dim), e.q.device))
            self._q_step += 1
            self._p_step += 1
            cfg = e.make_cfg(gamma=self.gamma, tau=self.tau, alpha=self.alpha, q_lr=self.learning_rate,
                             policy_lr=self.learning_rate, q_step=self._q_step, policy_step=self._p_step)
            items.append(dict(cfg=cfg, obs=s_obs, actions=s_actions, rewards=s_rewards, next_obs=s_next_obs, dones=s_dones, w=w,
                              eps_next=eps[0], eps_pi=eps[1]))
            self._n_updates += 1
        if len(items) == 1:
            it = items[0]
            self._out = e.update(it.pop("cfg"), **it)
        else:
            self._out = e.update_n(items)[-1]
        if self.log and self.global_step % 100 == 0:
            import wandb
            wandb.log({"losses/critic_loss": float(self._out["critic_loss"][0].item()),
                       "losses/policy_loss": float(self._out["policy_loss"][0].item()),
                       "global_step": self.global_step})

    def last_losses(self):
        """(critic_loss, policy_loss) of the most recent gradient step as host floats (synchronises)."""
        return float(self._out["critic_loss"][0].item()), float(self._out["policy_loss"][0].item())

    @th.no_grad()
    def eval(self, obs: Union[np.ndarray, th.Tensor], w: Union[np.ndarray, th.Tensor], torch_action=False):
        """``capql.py:364-377``: the deterministic action ``tanh(mean) * scale + bias``."""
        obs = as_f32(obs, self.engine.q.device).reshape(1, -1)
        w = as_f32(w, self.engine.q.device).reshape(1, -1)
        action = self.engine.policy_forward(obs, w)[0, 0]
        return action if torch_action else action.detach().cpu().numpy()

    @th.no_grad()
    def eval_batch(self, obs: np.ndarray, w: np.ndarray) -> np.ndarray:
        """``eval`` for n (observation, weight) pairs per pass (lock-step evaluation episodes, ``evaluation.py``)."""
        e = self.engine
        obs = as_f32(np.asarray(obs, dtype=np.float32), e.q.device).reshape(-1, e.D)
        w = as_f32(np.asarray(w, dtype=np.float32), e.q.device).reshape(-1, e.R)
        out = [e.policy_forward(obs[b:b + e.max_rows].contiguous(), w[b:b + e.max_rows].contiguous())[0]
               for b in range(0, obs.shape[0], e.max_rows)]
        return th.cat(out, dim=0).cpu().numpy()

    def train(self, total_timesteps: int, eval_env=None, ref_point: Optional[np.ndarray] = None,
              known_pareto_front: Optional[List[np.ndarray]] = None, num_eval_weights_for_front: int = 100,
              num_eval_episodes_for_front: int = 5, num_eval_weights_for_eval: int = 50, eval_freq: int = 10000,
              reset_num_timesteps: bool = False, checkpoints: bool = False, save_freq: int = 10000):
        """``capql.py:379-484``."""
        eval_weights = None
        if self.log:
            self.register_additional_config({
                "total_timesteps": total_timesteps, "ref_point": ref_point.tolist(), "known_front": known_pareto_front,
                "num_eval_weights_for_front": num_eval_weights_for_front,
                "num_eval_episodes_for_front": num_eval_episodes_for_front,
                "num_eval_weights_for_eval": num_eval_weights_for_eval, "eval_freq": eval_freq,
                "reset_num_timesteps": reset_num_timesteps})
            from morl_baselines.common.evaluation import log_all_multi_policy_metrics
            from morl_baselines.common.weights import equally_spaced_weights
            eval_weights = equally_spaced_weights(self.reward_dim, n=num_eval_weights_for_front)
        angle = th.pi * (22.5 / 180)
        weight_sampler = WeightSamplerAngle(self.env.unwrapped.reward_dim, angle)
        self.global_step = 0 if reset_num_timesteps else self.global_step
        self.num_episodes = 0 if reset_num_timesteps else self.num_episodes
        obs, info = self.env.reset()
        for _ in range(1, total_timesteps + 1):
            self.global_step += 1
            tensor_w = weight_sampler.sample(1).view(-1)
            w = tensor_w.detach().cpu().numpy()
            if self.global_step < self.learning_starts:
                action = self.env.action_space.sample()
            else:
                action = self.eval(obs, w)
            next_obs, vector_reward, terminated, truncated, info = self.env.step(action)
            self.replay_buffer.push(obs, action, w, vector_reward, next_obs, terminated)
            if self.global_step >= self.learning_starts:
                self.update()
            if terminated or truncated:
                obs, _ = self.env.reset()
                self.num_episodes += 1
            else:
                obs = next_obs
            if self.log and self.global_step % eval_freq == 0:
                returns_test_tasks = front_returns(self, eval_env, eval_weights, rep=num_eval_episodes_for_front)
                log_all_multi_policy_metrics(current_front=returns_test_tasks, hv_ref_point=ref_point,
                                             reward_dim=self.reward_dim, global_step=self.global_step,
                                             n_sample_weights=num_eval_weights_for_eval, ref_front=known_pareto_front)
            if checkpoints and self.global_step % save_freq == 0:
                self.save(filename=f"CAPQL step={self.global_step}", save_replay_buffer=False)
        self.close_wandb()
