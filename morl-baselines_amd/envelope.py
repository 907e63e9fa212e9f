"""Envelope Q-learning with the update path on MI355X HIP kernels.

Drop-in for the reference ``Envelope`` (``multi_policy/envelope/envelope.py:80-572``): same constructor keyword
arguments, ``update()``, ``envelope_target()``, ``ddqn_target()``, ``max_action()``, ``act()``, ``eval()``,
``train()``, ``save()`` / ``load()`` (same checkpoint keys), ``get_config()``.  What differs is where the arithmetic
runs: ``update()`` gathers the batch, evaluates the de-duplicated B*W next-state rows, takes the envelope arg-max,
and does loss / backward / clip / Adam entirely in ``libmorl_hip.so`` (one C call per gradient step); nothing is
synchronised with the host unless logging asks for a scalar.  There is no CPU fallback.
"""
from __future__ import annotations

import os
import re
from typing import List, Optional, Union

import numpy as np
import torch as th
import torch.optim as optim

from . import ops
from .evaluation import front_returns
from .api import MOAgent, MOPolicy, reference_method
from .native import NativeLib, load_library
from .qnet import QNet
from .replay import PrioritizedReplayBuffer, ReplayBuffer


def random_weights(dim: int, n: int = 1, dist: str = "dirichlet", seed=None, rng=None) -> np.ndarray:
    """``common/weights.py:10-35`` (host RNG; W x R floats -- stays on the host)."""
    if rng is None:
        rng = np.random.default_rng(seed)
    if dist == "gaussian":
        w = np.abs(rng.standard_normal((n, dim)))
        w = w / np.add.reduce(w, axis=1, keepdims=True)     # (= np.linalg.norm(w, ord=1, axis=1, keepdims=True): the same reduction)
    elif dist == "dirichlet":
        w = rng.dirichlet(np.ones(dim), n)
    else:
        raise ValueError(f"Unknown distribution {dist}")
    return w[0] if n == 1 else w


def linearly_decaying_value(initial_value, decay_period, step, warmup_steps, final_value):
    """``common/utils.py:10-32``."""
    steps_left = decay_period + warmup_steps - step
    bonus = (initial_value - final_value) * steps_left / decay_period
    value = final_value + bonus
    return np.clip(value, min(initial_value, final_value), max(initial_value, final_value))


class Envelope(MOPolicy, MOAgent):
    """Envelope Q-Learning (Yang et al. 2019) -- HIP update path."""

    def __init__(
        self,
        env,
        learning_rate: float = 3e-4,
        initial_epsilon: float = 0.01,
        final_epsilon: float = 0.01,
        epsilon_decay_steps: int = None,
        tau: float = 1.0,
        target_net_update_freq: int = 200,
        buffer_size: int = int(1e6),
        net_arch: List = [256, 256, 256, 256],
        batch_size: int = 256,
        learning_starts: int = 100,
        gradient_updates: int = 1,
        gamma: float = 0.99,
        max_grad_norm: Optional[float] = 1.0,
        envelope: bool = True,
        num_sample_w: int = 4,
        per: bool = True,
        per_alpha: float = 0.6,
        initial_homotopy_lambda: float = 0.0,
        final_homotopy_lambda: float = 1.0,
        homotopy_decay_steps: int = None,
        project_name: str = "MORL-Baselines",
        experiment_name: str = "Envelope",
        wandb_entity: Optional[str] = None,
        log: bool = True,
        seed: Optional[int] = None,
        device: Union[th.device, str] = "auto",
        group: Optional[str] = None,
        lib: Optional[NativeLib] = None,
        engine: Optional[int] = None,
    ):
        MOAgent.__init__(self, env, device=device, seed=seed)
        MOPolicy.__init__(self, device=device)
        self.device = th.device(self.device)
        self.lib = lib or load_library()
        self.lib.check_device(th.empty(0, device=self.device))   # loud failure on a CPU device with the gfx950 build
        self.learning_rate = learning_rate
        self.initial_epsilon = initial_epsilon
        self.epsilon = initial_epsilon
        self.epsilon_decay_steps = epsilon_decay_steps
        self.final_epsilon = final_epsilon
        self.tau = tau
        self.target_net_update_freq = target_net_update_freq
        self.gamma = gamma
        self.max_grad_norm = max_grad_norm
        self.buffer_size = buffer_size
        self.net_arch = net_arch
        self.learning_starts = learning_starts
        self.batch_size = batch_size
        self.per = per
        self.per_alpha = per_alpha
        self.gradient_updates = gradient_updates
        self.initial_homotopy_lambda = initial_homotopy_lambda
        self.final_homotopy_lambda = final_homotopy_lambda
        self.homotopy_decay_steps = homotopy_decay_steps
        self.envelope = envelope
        self.num_sample_w = num_sample_w
        self.homotopy_lambda = self.initial_homotopy_lambda
        self.experiment_name = experiment_name

        self.q_net = QNet(self.observation_shape, self.action_dim, self.reward_dim, net_arch=net_arch,
                          device=self.device, lib=self.lib, max_batch=batch_size, max_weights=num_sample_w,
                          engine=engine)
        self.target_q_net = QNet(self.observation_shape, self.action_dim, self.reward_dim, net_arch=net_arch,
                                 device=self.device, lib=self.lib, max_batch=1, max_weights=1, engine=engine)
        self.target_q_net.load_state_dict(self.q_net.state_dict())
        for param in self.target_q_net.parameters():
            param.requires_grad = False

        # torch's Adam object is kept for state_dict()/load_state_dict() compatibility; its state tensors are views of
        # the flat buffers the HIP kernel updates (torch/optim/adam.py state layout: step, exp_avg, exp_avg_sq)
        self.q_optim = optim.Adam(self.q_net.parameters(), lr=self.learning_rate)
        P = self.q_net.ctx.n_params
        self._grads = th.zeros(P, dtype=th.float32, device=self.device)
        self._exp_avg = th.zeros(P, dtype=th.float32, device=self.device)
        self._exp_avg_sq = th.zeros(P, dtype=th.float32, device=self.device)
        self._adam_step = 0
        self._bind_optimizer_state()

        if self.per:
            self.replay_buffer = PrioritizedReplayBuffer(self.observation_shape, 1, rew_dim=self.reward_dim,
                                                         max_size=buffer_size, action_dtype=np.uint8,
                                                         device=self.device, lib=self.lib)
        else:
            self.replay_buffer = ReplayBuffer(self.observation_shape, 1, rew_dim=self.reward_dim, max_size=buffer_size,
                                              action_dtype=np.uint8, device=self.device, lib=self.lib)
        self._out = None            # device scalars / vectors of the last gradient step
        self._losses: List[th.Tensor] = []
        self.log = log
        if log:
            self.setup_wandb(project_name, experiment_name, wandb_entity, group)

    # ------------------------------------------------------------------------------------------------------------
    def _bind_optimizer_state(self) -> None:
        """Point every parameter's .grad and Adam state at views of the flat buffers (torch-visible, kernel-owned)."""
        params = self.q_net.ordered_parameters()
        slices = []
        for (wo, wshape, bo, bn) in self.q_net.ctx.layer_slices():
            slices += [(wo, wshape[0] * wshape[1], wshape), (bo, bn, (bn,))]
        for prm, (off, n, shape) in zip(params, slices):
            prm.grad = self._grads[off:off + n].view(shape)
            self.q_optim.state[prm] = {
                "step": th.tensor(float(self._adam_step)),
                "exp_avg": self._exp_avg[off:off + n].view(shape),
                "exp_avg_sq": self._exp_avg_sq[off:off + n].view(shape),
            }

    def _sync_optimizer_step(self) -> None:
        for st in self.q_optim.state.values():
            st["step"].fill_(float(self._adam_step))

    def get_config(self):
        return {
            "env_id": self.env.unwrapped.spec.id,
            "learning_rate": self.learning_rate,
            "initial_epsilon": self.initial_epsilon,
            "epsilon_decay_steps": self.epsilon_decay_steps,
            "batch_size": self.batch_size,
            "tau": self.tau,
            "clip_grand_norm": self.max_grad_norm,
            "target_net_update_freq": self.target_net_update_freq,
            "gamma": self.gamma,
            "use_envelope": self.envelope,
            "num_sample_w": self.num_sample_w,
            "net_arch": self.net_arch,
            "per": self.per,
            "gradient_updates": self.gradient_updates,
            "buffer_size": self.buffer_size,
            "initial_homotopy_lambda": self.initial_homotopy_lambda,
            "final_homotopy_lambda": self.final_homotopy_lambda,
            "homotopy_decay_steps": self.homotopy_decay_steps,
            "learning_starts": self.learning_starts,
            "seed": self.seed,
        }

    # -- checkpointing: same keys as envelope.py:230-261 -----------------------------------------------------------
    def save(self, save_replay_buffer: bool = True, save_dir: str = "weights/", filename: Optional[str] = None):
        if not os.path.isdir(save_dir):
            os.makedirs(save_dir)
        self._sync_optimizer_step()
        saved_params = {"q_net_state_dict": self.q_net.state_dict(),
                        "q_net_optimizer_state_dict": self.q_optim.state_dict()}
        if save_replay_buffer:
            saved_params["replay_buffer"] = self.replay_buffer
        filename = self.experiment_name if filename is None else filename
        th.save(saved_params, save_dir + "/" + filename + ".tar")

    def load(self, path: str, load_replay_buffer: bool = True):
        params = th.load(path, weights_only=False)
        self.q_net.load_state_dict(params["q_net_state_dict"])
        self.target_q_net.load_state_dict(params["q_net_state_dict"])   # envelope.py:257-258
        self.q_net.ctx.invalidate_shadows()                             # (in-place writes behind the library's back)
        self.q_optim.load_state_dict(params["q_net_optimizer_state_dict"])
        # load_state_dict re-allocates the state tensors: copy them back into the flat buffers and re-bind the views
        prms = self.q_net.ordered_parameters()
        step = 0
        with th.no_grad():
            for prm in prms:
                st = self.q_optim.state.get(prm)
                if st:
                    step = int(float(st["step"]))
            loaded = [(self.q_optim.state.get(prm) or {}) for prm in prms]
        self._adam_step = step
        off = 0
        for prm, st in zip(prms, loaded):
            n = prm.numel()
            if "exp_avg" in st:
                self._exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self._exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            off += n
        self._bind_optimizer_state()
        if load_replay_buffer and "replay_buffer" in params:
            self.replay_buffer = params["replay_buffer"]

    # -- the hot path ----------------------------------------------------------------------------------------------
    def __sample_batch_experiences(self, aux=None):
        # (the same launch makes the K-major shadow weights of both networks for the step that follows)
        return self.replay_buffer.sample(self.batch_size, to_tensor=True, device=self.device, aux=aux,
                                         prepare=(self.q_net.ctx, self.q_net.flat, self.target_q_net.flat))

    def _draw_weights(self):
        """The step's ``num_sample_w`` weight vectors (``envelope.py:281-283``: host RNG, fp32): staged in pinned memory and
        moved to a persistent device buffer by the batch-gather launch (no copy launch of their own).  Returns the ``aux``
        argument of ``replay_buffer.sample`` and the device tensor (valid once that launch has been enqueued)."""
        W, R = self.num_sample_w, self.reward_dim
        ring = self._w_ring = ops.HostRing.fit(getattr(self, "_w_ring", None), self.lib, self.device, W * R, th.float32)
        if getattr(self, "_w_dev", None) is None or self._w_dev.shape != (W, R):
            self._w_dev = th.zeros((W, R), dtype=th.float32, device=self.device)
        slot, ptr = ring.next(W * R)
        slot[:] = random_weights(dim=R, n=W, dist="gaussian", rng=self.np_random).reshape(-1)   # float64 -> fp32 (.float())
        return (ptr, self._w_dev), self._w_dev

    def _step_block(self, n: int):
        """The persistent argument block of ``morl_envelope_update_n`` (``native.StepIO``) with the tensors it points at: the batch
        the gather launch writes and the update reads, the device copy of the sampled weights, the loss / norm / priority outputs.
        Built once, rebuilt when anything it captured changes (batch size, weight count, a re-bound buffer, a loaded replay
        buffer, a hyper-parameter); ``None`` when the replay buffer is not one of ``replay.py``'s device-mirrored ones."""
        buf = self.replay_buffer
        if not isinstance(buf, ReplayBuffer) or buf._Ad != 1 or not buf._int_actions:
            return None
        key = (self.batch_size, self.num_sample_w, self.q_net.flat.data_ptr(), self.target_q_net.flat.data_ptr(),
               self._grads.data_ptr(), self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr(), id(buf), buf.records.data_ptr(),
               buf.tree_dev.data_ptr() if hasattr(buf, "tree_dev") else 0,
               self.gamma, self.learning_rate, self.max_grad_norm, self.envelope, self.per_alpha, self.q_net.ctx.handle)
        st = self.__dict__.get("_step_state")
        if st is not None and st.key == key and st.n_alloc >= n:
            return st
        import ctypes as C
        import types
        from .native import StepIO
        B, W, R, D, dev = self.batch_size, self.num_sample_w, self.reward_dim, self.observation_dim, self.device
        self.lib.check_device(self.q_net.flat, self.target_q_net.flat, self._grads, self._exp_avg, self._exp_avg_sq, buf.records)
        per = isinstance(buf, PrioritizedReplayBuffer)
        # (st.buf: the block holds raw addresses of the buffer's tensors -- the reference keeps the buffer, and with it the identity
        # id(buf) in the key, alive for as long as the block is)
        st = types.SimpleNamespace(key=key, n_alloc=max(n, 1), per=per, buf=buf)
        f32 = dict(dtype=th.float32, device=dev)
        st.obs, st.next_obs = th.empty((B, D), **f32), th.empty((B, D), **f32)
        st.rewards, st.dones = th.empty((B, R), **f32), th.empty((B, 1), **f32)
        st.actions = th.empty((B,), dtype=th.int32, device=dev)
        st.idx = th.empty((B,), dtype=th.int64, device=dev)
        st.weights = th.zeros((W, R), **f32)
        st.loss, st.grad_norm = th.zeros((st.n_alloc,), **f32), th.zeros((st.n_alloc,), **f32)
        st.priority = th.zeros((B,), **f32)
        st.loss_views = [st.loss[k] for k in range(st.n_alloc)]
        st.outs = [{"loss": st.loss[k], "grad_norm": st.grad_norm[k], "priority": st.priority} for k in range(st.n_alloc)]
        io = StepIO()
        io.params_online, io.params_target = self.q_net.flat.data_ptr(), self.target_q_net.flat.data_ptr()
        io.grads, io.exp_avg, io.exp_avg_sq = self._grads.data_ptr(), self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr()
        if per:
            self.lib.check_device(buf.tree_dev, buf.running_max)
            io.tree, io.running_max, io.n_levels = buf.tree_dev.data_ptr(), buf.running_max.data_ptr(), buf.n_levels
        io.records, io.capacity, io.record_floats = buf.records.data_ptr(), buf.records.shape[0], buf.records.shape[1]
        io.obs, io.next_obs, io.rewards, io.dones = (t.data_ptr() for t in (st.obs, st.next_obs, st.rewards, st.dones))
        io.actions, io.idx, io.weights = st.actions.data_ptr(), st.idx.data_ptr(), st.weights.data_ptr()
        io.D, io.R, io.B, io.W = D, R, B, W
        io.cfg = ops._update_cfg(self.gamma, self.learning_rate, 1, self.max_grad_norm, 0.0, self.envelope, 0.9, 0.999, 1e-8, True)
        io.cfg.per_alpha = float(self.per_alpha)
        st.io, st.io_ref = io, C.byref(io)
        st.loss_ptr, st.gn_ptr, st.prio_ptr = st.loss.data_ptr(), st.grad_norm.data_ptr(), st.priority.data_ptr()
        st.fn = self.lib.lib.morl_envelope_update_n
        self._step_state = st
        return st

    def update(self):
        """``envelope.py:267-367``: the whole ``gradient_updates`` loop is ONE library call (``morl_envelope_update_n``), nothing is
        synchronised with the host.  The host only draws what the reference draws -- per generator in the reference's order: the
        batch's uniforms / indices from the global numpy RNG (``prioritized_buffer.py:40`` / ``buffer.py:82``), the sampled
        weights from ``self.np_random`` (``:279-283``) -- into pinned slots the gather launches read in place."""
        n = self.gradient_updates
        self.q_net.ensure_capacity(self.batch_size, self.num_sample_w)
        # (a prioritised batch larger than one tree-update launch holds -- 1 024 entries -- takes the call-by-call path, whose
        # priority update goes through update_priorities' ascending blocks)
        big_per = self.per and self.batch_size > PrioritizedReplayBuffer.TREE_BLOCK
        st = None if big_per else self._step_block(n)
        if st is None:
            return self._update_by_calls()
        W, R = self.num_sample_w, self.reward_dim
        WR = W * R
        ring = self._w_ring = ops.HostRing.fit(self.__dict__.get("_w_ring"), self.lib, self.device, n * WR, th.float32)
        slot, w_ptr = ring.next(n * WR)
        if n == 1:
            slot[:] = random_weights(dim=R, n=W, dist="gaussian", rng=self.np_random).reshape(-1)    # float64 -> fp32 (.float())
        else:
            for k in range(n):
                slot[k * WR:(k + 1) * WR] = random_weights(dim=R, n=W, dist="gaussian", rng=self.np_random).reshape(-1)
        u_ptr, i_ptr = self.replay_buffer.draw_batches(self.batch_size, n)
        a0 = self._adam_step + 1
        self._adam_step += n
        rc = st.fn(self.q_net.ctx.handle, st.io_ref, n, u_ptr, i_ptr, w_ptr, a0, float(self.homotopy_lambda), st.loss_ptr, st.gn_ptr,
                   st.prio_ptr, self.lib.stream_of(st.loss))
        # (the pinned slots are handed out again only behind an event recorded NOW: kernels of the iterations that were enqueued
        # before a failure still read them)
        ring.mark_used()
        self.replay_buffer.mark_drawn()
        if rc:
            # "update k of n: ...": k optimiser steps were enqueued before the failing iteration -- Adam's step count keeps them
            msg = self.lib.lib.morl_last_error().decode()
            m = re.match(r"update (\d+) of \d+:", msg)
            self._adam_step -= n - (min(int(m.group(1)), n) if m else 0)
            self.lib.check(rc)
        self._losses = st.loss_views[:n]
        self._out = st.outs[n - 1]
        self._finish_update(st.priority if st.per else None)

    def _update_by_calls(self):
        """The same loop as one ``sample`` + one ``morl_envelope_update`` call per iteration (foreign replay buffers; the A/B leg of
        the tests: bit-identical to ``update``)."""
        self._losses = []
        priority = None
        self.q_net.ensure_capacity(self.batch_size, self.num_sample_w)
        in_step = self.per and self.batch_size <= PrioritizedReplayBuffer.TREE_BLOCK
        for _ in range(self.gradient_updates):
            aux, sampled_w = self._draw_weights()
            b_obs, b_actions, b_rewards, b_next_obs, b_dones, b_inds = self.__sample_batch_experiences(aux)
            self._w_ring.mark_used()
            self._adam_step += 1
            self._out = ops.envelope_update(
                self.q_net.ctx, self.q_net.flat, self.target_q_net.flat, self._grads, self._exp_avg, self._exp_avg_sq,
                b_obs, b_next_obs, b_actions.reshape(-1).to(th.int32), b_rewards, b_dones.reshape(-1), sampled_w,
                gamma=self.gamma, lr=self.learning_rate, adam_step=self._adam_step, max_grad_norm=self.max_grad_norm,
                homotopy_lambda=float(self.homotopy_lambda), envelope=self.envelope, outputs=None,
                per=self.replay_buffer.per_update_args(b_inds, self.per_alpha) if in_step else None)
            self._losses.append(self._out["loss"])
            if self.per:
                priority = self._out["priority"]      # (the sum tree was updated inside the step: envelope.py:329-334)
                if not in_step:
                    # envelope.py:329-334 for a batch of more than 1 024 transitions: priority = (|td . w| + min_priority) ** alpha
                    # with the running maximum as it stood BEFORE this update (fp32, as numpy evaluates it), then update_priorities
                    # (ascending blocks: the tree of one batch_set, bit for bit)
                    rmax = self.replay_buffer.running_max.to(th.float32)
                    self.replay_buffer.update_priorities(b_inds, (priority + rmax).pow(th.tensor(self.per_alpha, dtype=th.float32, device=priority.device)))

        self._finish_update(priority)

    def _finish_update(self, priority=None):
        """Tail of ``Envelope.update`` (``envelope.py:336-367``): target sync, epsilon / homotopy schedules, logging.
        Shared by the single-GPU step and the weight-sharded step of ``distributed.py``."""
        if self.tau != 1 or self.global_step % self.target_net_update_freq == 0:
            ops.polyak(self.lib, self.q_net.flat, self.target_q_net.flat, self.tau)

        if self.epsilon_decay_steps is not None:
            self.epsilon = linearly_decaying_value(self.initial_epsilon, self.epsilon_decay_steps, self.global_step,
                                                   self.learning_starts, self.final_epsilon)
        if self.homotopy_decay_steps is not None:
            self.homotopy_lambda = linearly_decaying_value(self.initial_homotopy_lambda, self.homotopy_decay_steps,
                                                           self.global_step, self.learning_starts,
                                                           self.final_homotopy_lambda)
        if self.log and self.global_step % 100 == 0:
            import wandb
            log = {
                "losses/critic_loss": float(th.stack(self._losses).mean().item()),
                "metrics/epsilon": self.epsilon,
                "metrics/homotopy_lambda": self.homotopy_lambda,
                "global_step": self.global_step,
            }
            if self._out is not None and "grad_norm" in self._out:
                log["losses/grad_norm"] = float(self._out["grad_norm"].item())
            wandb.log(log)
            if self.per and priority is not None:
                wandb.log({"metrics/mean_priority": float(priority.mean().item())})

    def last_loss(self) -> float:
        """Host value of the most recent critic loss (synchronises)."""
        return float(self._losses[-1].item())

    # -- targets with the reference's signatures (envelope.py:404-463) -----------------------------------------------
    @th.no_grad()
    def envelope_target(self, obs: th.Tensor, w: th.Tensor, sampled_w: th.Tensor) -> th.Tensor:
        """obs (n, D), w (n, R): one scalarisation vector per row; sampled_w (W, R).  Returns (n, R)."""
        obs = obs.to(self.device, th.float32).contiguous()
        w = w.to(self.device, th.float32).contiguous()
        sampled_w = sampled_w.to(self.device, th.float32).contiguous()
        n, W = obs.size(0), sampled_w.size(0)
        self.q_net.ensure_capacity(n, W)
        ctx = self.q_net.ctx
        qo = ops.qnet_forward(ctx, self.q_net.flat, obs, sampled_w, row_order=0).view(n, W, self.action_dim, self.reward_dim)
        qt = ops.qnet_forward(ctx, self.target_q_net.flat, obs, sampled_w, row_order=0).view(n, W, self.action_dim, self.reward_dim)
        target, _, _ = ops.envelope_reduce_rows(self.lib, qo, qt, w)
        return target

    @th.no_grad()
    def ddqn_target(self, obs: th.Tensor, w: th.Tensor) -> th.Tensor:
        obs = obs.to(self.device, th.float32).contiguous()
        w = w.to(self.device, th.float32).contiguous()
        n = obs.size(0)
        self.q_net.ensure_capacity(n, 1)
        qo = ops.qnet_forward_rows(self.q_net.ctx, self.q_net.flat, obs, w).view(n, 1, self.action_dim, self.reward_dim)
        qt = ops.qnet_forward_rows(self.q_net.ctx, self.target_q_net.flat, obs, w).view(n, 1, self.action_dim, self.reward_dim)
        target, _, _ = ops.envelope_reduce_rows(self.lib, qo, qt, w)
        return target

    # -- acting ------------------------------------------------------------------------------------------------------
    def eval(self, obs: np.ndarray, w: np.ndarray) -> int:
        obs = th.as_tensor(obs).float().to(self.device)
        w = th.as_tensor(w).float().to(self.device)
        return self.max_action(obs, w)

    @th.no_grad()
    def eval_batch(self, obs: np.ndarray, w: np.ndarray) -> np.ndarray:
        """``eval`` for n (observation, weight) pairs in one pass (lock-step evaluation episodes, ``evaluation.py``):
        rows through the HIP forward, then the envelope kernel's arg-max with one candidate weight per row."""
        obs = th.as_tensor(np.asarray(obs)).float().to(self.device).reshape(-1, self.observation_dim).contiguous()
        w = th.as_tensor(np.asarray(w)).float().to(self.device).reshape(-1, self.reward_dim).contiguous()
        n = obs.size(0)
        self.q_net.ensure_capacity(n, 1)
        q = ops.qnet_forward_rows(self.q_net.ctx, self.q_net.flat, obs, w).view(n, 1, self.action_dim, self.reward_dim)
        _, _, ac = ops.envelope_reduce_rows(self.lib, q, q, w)
        return ac.cpu().numpy().astype(np.int64)

    def act(self, obs: th.Tensor, w: th.Tensor) -> int:
        if self.np_random.random() < self.epsilon:
            return self.env.action_space.sample()
        return self.max_action(obs, w)

    @th.no_grad()
    def max_action(self, obs: th.Tensor, w: th.Tensor) -> int:
        """``envelope.py:389-402`` in one C call (``morl_envelope_greedy_actions``): the row through the HIP forward, fma-chain
        scalarisation (what the reference's unbatched einsum evaluates to) and first arg-max on the device; one 4-byte
        read-back for the environment."""
        obs = th.as_tensor(obs).to(self.device, th.float32).reshape(1, -1).contiguous()
        w = th.as_tensor(w).to(self.device, th.float32).reshape(1, -1).contiguous()
        self.q_net.ensure_capacity(1, 1)
        return int(ops.envelope_greedy_actions(self.q_net.ctx, self.q_net.flat, obs, w).item())

    # -- training loop (envelope.py:465-572) ---------------------------------------------------------------------------
    def train(self, total_timesteps: int, eval_env=None, ref_point: Optional[np.ndarray] = None,
              known_pareto_front: Optional[List[np.ndarray]] = None, weight: Optional[np.ndarray] = None,
              total_episodes: Optional[int] = None, reset_num_timesteps: bool = True, eval_freq: int = 10000,
              num_eval_weights_for_front: int = 100, num_eval_episodes_for_front: int = 5,
              num_eval_weights_for_eval: int = 50, reset_learning_starts: bool = False, verbose: bool = False):
        """``envelope.py:465-575``.  With ``morl_baselines`` importable this IS the reference's method (``api.reference_method``): its
        loop calls this class's ``act`` / ``update`` / replay buffer, everything else is the reference's own code.  The loop below is
        the restatement for machines without the reference (same draws from the same generators in the same order: the seeded traces
        of tests/test_train_traces.py), plus one extension: a LIST of evaluation environments is rolled out in lock-step."""
        ref_train = None if isinstance(eval_env, (list, tuple)) else reference_method(
            "morl_baselines.multi_policy.envelope.envelope", "Envelope", "train")
        if ref_train is not None:
            return ref_train(self, total_timesteps, eval_env=eval_env, ref_point=ref_point, known_pareto_front=known_pareto_front,
                             weight=weight, total_episodes=total_episodes, reset_num_timesteps=reset_num_timesteps, eval_freq=eval_freq,
                             num_eval_weights_for_front=num_eval_weights_for_front, num_eval_episodes_for_front=num_eval_episodes_for_front,
                             num_eval_weights_for_eval=num_eval_weights_for_eval, reset_learning_starts=reset_learning_starts,
                             verbose=verbose)
        if eval_env is not None:
            assert ref_point is not None, "Reference point must be provided for the hypervolume computation."
        self.global_step = 0 if reset_num_timesteps else self.global_step
        self.num_episodes = 0 if reset_num_timesteps else self.num_episodes
        if reset_learning_starts:
            self.learning_starts = self.global_step
        num_episodes = 0
        eval_weights = None
        if eval_env is not None and self.log:
            from morl_baselines.common.evaluation import log_all_multi_policy_metrics  # reference, unchanged
            from morl_baselines.common.weights import equally_spaced_weights
            eval_weights = equally_spaced_weights(self.reward_dim, n=num_eval_weights_for_front)
        obs, _ = self.env.reset()
        w = weight if weight is not None else random_weights(self.reward_dim, 1, dist="gaussian", rng=self.np_random)
        tensor_w = th.tensor(w).float().to(self.device)
        for _ in range(1, total_timesteps + 1):
            if total_episodes is not None and num_episodes == total_episodes:
                break
            if self.global_step < self.learning_starts:
                action = self.env.action_space.sample()
            else:
                action = self.act(th.as_tensor(obs).float().to(self.device), tensor_w)
            next_obs, vec_reward, terminated, truncated, info = self.env.step(action)
            self.global_step += 1
            self.replay_buffer.add(obs, action, vec_reward, next_obs, terminated)
            if self.global_step >= self.learning_starts:
                self.update()
            if eval_weights is not None and self.global_step % eval_freq == 0:
                if isinstance(eval_env, (list, tuple)):      # one environment per weight: lock-step roll-outs (evaluation.py)
                    current_front = front_returns(self, eval_env, eval_weights, rep=num_eval_episodes_for_front)
                else:
                    current_front = [self.policy_eval(eval_env, weights=ew, num_episodes=num_eval_episodes_for_front,
                                                      log=self.log)[3] for ew in eval_weights]
                log_all_multi_policy_metrics(current_front=current_front, hv_ref_point=ref_point,
                                             reward_dim=self.reward_dim, global_step=self.global_step,
                                             n_sample_weights=num_eval_weights_for_eval, ref_front=known_pareto_front)
            if terminated or truncated:
                obs, _ = self.env.reset()
                num_episodes += 1
                self.num_episodes += 1
                if weight is None:
                    w = random_weights(self.reward_dim, 1, dist="gaussian", rng=self.np_random)
                    tensor_w = th.tensor(w).float().to(self.device)
            else:
                obs = next_obs


EnvelopeHIP = Envelope
