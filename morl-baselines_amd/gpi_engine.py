"""Device-resident ensemble of GPI-PD Q-networks (discrete actions) + one-call update on ``morl_gpi_*`` of the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch as th

from . import native
from .native import GPIBatch, GPICfg, GPIDesc, GPIOut, GPIPer, NativeLib


class GPIEngine:
    def __init__(self, obs_dim: int, n_actions: int, reward_dim: int, net_arch: Sequence[int], *, max_rows: int,
                 max_support: int = 8, num_nets: int = 2, layer_norm: bool = True, drop_rate: float = 0.01,
                 device="cuda", lib: Optional[NativeLib] = None):
        self.lib = lib or native.load_library()
        self.device = th.device(device)
        if len(net_arch) < 2 or len(net_arch) > native.MORL_MAX_LAYERS:
            raise ValueError(f"net_arch needs 2..{native.MORL_MAX_LAYERS} entries")
        d = GPIDesc()
        d.obs_dim, d.reward_dim, d.n_actions, d.n_hidden = obs_dim, reward_dim, n_actions, len(net_arch)
        for i, h in enumerate(net_arch):
            d.hidden[i] = int(h)
        d.num_nets, d.layer_norm, d.drop_rate = num_nets, int(bool(layer_norm)), float(drop_rate)
        d.max_rows, d.max_support = max_rows, max_support
        self.desc = d
        self.D, self.A, self.R, self.arch = obs_dim, n_actions, reward_dim, [int(h) for h in net_arch]
        self.num_nets, self.layer_norm, self.drop_rate = num_nets, bool(layer_norm), float(drop_rate)
        self.max_rows, self.max_support = max_rows, max_support
        self.P = int(self.lib.lib.morl_gpi_param_count(C.byref(d)))
        if self.P < 0:
            self.lib.check(-1)
        h = C.c_void_p()
        self.lib.check(self.lib.lib.morl_gpi_create(C.byref(h), C.byref(d)))
        self._h = h.value
        z = lambda: th.zeros((num_nets, self.P), dtype=th.float32, device=self.device)  # noqa: E731
        self.q, self.q_target, self.exp_avg, self.exp_avg_sq = z(), z(), z(), z()
        self.lib.check_device(self.q)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.lib.morl_gpi_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def shapes(self):
        """``QNet.parameters()`` order: weights_features, state_features, net."""
        h0 = self.arch[0]
        out, d = [(h0, self.R), (h0,), (h0, self.D), (h0,)], h0
        for h in self.arch[1:]:
            out += [(h, d), (h,)]
            if self.layer_norm:
                out += [(h,), (h,)]
            d = h
        return out + [(self.A * self.R, d), (self.A * self.R,)]

    def views(self, buf: th.Tensor, n: int) -> List[th.Tensor]:
        out, o = [], 0
        for s in self.shapes():
            k = int(np.prod(s))
            out.append(buf[n, o:o + k].view(s))
            o += k
        assert o == self.P
        return out

    def _f32(self, t) -> th.Tensor:
        t = th.as_tensor(t)
        if t.dtype != th.float32 or t.device != self.q.device or not t.is_contiguous():
            t = t.to(self.q.device, th.float32).contiguous()
        return t

    def mask_bytes(self, rows: int) -> int:
        return int(self.lib.lib.morl_gpi_mask_bytes(C.byref(self.desc), rows))

    def _pack(self, *, obs, actions, rewards, next_obs, dones, w, sampled_w=None, gamma=0.99, lr=3e-4, adam_step=1,
              min_priority=0.01, max_grad_norm=None, gpi_pd=True, n_per=0, dropout_seed=0, apply_step=True,
              drop_masks: Optional[th.Tensor] = None, want: Sequence[str] = ("critic_loss",)):
        """The C structs of one ``morl_gpi_update``: (GPIBatch, GPICfg, GPIOut, {name: output tensor}, tensors to keep alive)."""
        obs, rewards, next_obs, w = self._f32(obs), self._f32(rewards), self._f32(next_obs), self._f32(w)
        dones = self._f32(dones).reshape(-1)
        rows = obs.shape[0]
        actions = th.as_tensor(actions).to(self.q.device).reshape(-1).to(th.int32).contiguous()
        K = 0
        if gpi_pd:
            sampled_w = self._f32(sampled_w).reshape(-1, self.R)
            K = sampled_w.shape[0]
        cfg = GPICfg()
        cfg.gamma, cfg.min_priority = gamma, min_priority
        cfg.max_grad_norm = -1.0 if max_grad_norm is None else float(max_grad_norm)
        cfg.lr, cfg.beta1, cfg.beta2, cfg.eps = lr, 0.9, 0.999, 1e-8
        cfg.adam_step, cfg.gpi_pd, cfg.n_per, cfg.apply_step, cfg.dropout_seed = adam_step, int(gpi_pd), n_per, \
            int(apply_step), dropout_seed
        shapes = dict(critic_loss=(1,), td_error=(max(n_per, 1),), gtd_error=(max(n_per, 1),), target_q=(rows, self.R),
                      target_q_envelope=(rows, self.R), grads=(self.num_nets, self.P), grad_norm=(self.num_nets,))
        out, res = GPIOut(), {}
        for name in want:
            res[name] = th.zeros(shapes[name], dtype=th.float32, device=self.q.device)
            setattr(out, name, res[name].data_ptr())
        if drop_masks is not None and (drop_masks.dtype != th.uint8 or not drop_masks.is_contiguous()):
            raise ValueError("drop_masks must be contiguous uint8")
        self.lib.check_device(obs, actions, rewards, next_obs, dones, w, sampled_w, drop_masks)
        b = GPIBatch()
        b.obs, b.actions, b.rewards, b.next_obs, b.dones, b.w = (t.data_ptr() for t in (obs, actions, rewards, next_obs, dones, w))
        b.sampled_w = None if sampled_w is None else sampled_w.data_ptr()
        b.drop_masks = None if drop_masks is None else drop_masks.data_ptr()
        b.rows, b.K = rows, K
        return b, cfg, out, res, (obs, actions, rewards, next_obs, dones, w, sampled_w, drop_masks)

    def update(self, **kw) -> Dict[str, th.Tensor]:
        """One ``morl_gpi_update`` (the loop body of ``GPIPD.update``, gpi_pd.py:418-520); keyword arguments: see ``_pack``."""
        b, cfg, out, res, _keep = self._pack(**kw)
        self.lib.check(self.lib.lib.morl_gpi_update(
            self._h, self.q.data_ptr(), self.q_target.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            b.obs, b.actions, b.rewards, b.next_obs, b.dones, b.w, b.rows, b.sampled_w, b.K, b.drop_masks, C.byref(cfg),
            C.byref(out), self.lib.stream_of(self.q)))
        return res

    def update_n(self, items: Sequence[dict]):
        """``morl_gpi_update_n``: the reference's ``for g in range(self.gradient_updates)`` loop in ONE library entry.  ``items``:
        the keyword arguments of ``update`` for each of the n updates (batches already drawn); returns their output dicts."""
        n = len(items)
        packed = [self._pack(**kw) for kw in items]
        bs, cs, os_ = (GPIBatch * n)(), (GPICfg * n)(), (GPIOut * n)()
        for k, (b, cfg, out, _res, _keep) in enumerate(packed):
            bs[k], cs[k], os_[k] = b, cfg, out
        self.lib.check(self.lib.lib.morl_gpi_update_n(
            self._h, self.q.data_ptr(), self.q_target.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), n, bs, cs,
            os_, self.lib.stream_of(self.q)))
        return [p[3] for p in packed]

    def update_n_per(self, items: Sequence[dict], *, buffer, u01: np.ndarray, doubled: bool, use_gtd: bool, alpha: float,
                     min_priority: float):
        """``morl_gpi_update_n_per``: the same loop with prioritised replay in ONE library entry -- iteration k samples
        ``buffer`` (a device ``PrioritizedReplayBuffer``) through its sum tree with the unit uniforms ``u01[k]``, gathers the
        transitions into the batch tensors of ``items[k]`` (obs / actions / rewards / next_obs / dones: scratch the caller
        allocated, B or 2 B rows; w / sampled_w filled), updates, and writes max(|td|, min_priority) ** alpha back into the
        tree.  Returns (output dicts, sampled indices [n][B])."""
        n = len(items)
        u01 = np.ascontiguousarray(u01, dtype=np.float64).reshape(n, -1)
        B = u01.shape[1]
        buffer.flush()
        packed = [self._pack(**kw) for kw in items]
        bs, cs, os_ = (GPIBatch * n)(), (GPICfg * n)(), (GPIOut * n)()
        for k, (b, cfg, out, _res, _keep) in enumerate(packed):
            bs[k], cs[k], os_[k] = b, cfg, out
        dev = self.q.device
        u_dev = th.as_tensor(u01).to(dev)
        idx = th.empty((n, B), dtype=th.int64, device=dev)
        per = GPIPer()
        per.tree, per.running_max, per.u01 = buffer.tree_dev.data_ptr(), buffer.running_max.data_ptr(), u_dev.data_ptr()
        per.records, per.idx = buffer.records.data_ptr(), idx.data_ptr()
        per.capacity, per.record_floats = buffer.records.shape[0], buffer.records.shape[1]
        per.n_levels, per.D, per.R, per.action_dim, per.B = buffer.n_levels, buffer._D, buffer._R, buffer._Ad, B
        per.doubled, per.use_gtd, per.alpha, per.min_priority = int(doubled), int(use_gtd), float(alpha), float(min_priority)
        self.lib.check_device(buffer.tree_dev, buffer.running_max, buffer.records)
        self.lib.check(self.lib.lib.morl_gpi_update_n_per(
            self._h, self.q.data_ptr(), self.q_target.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), n,
            C.byref(per), bs, cs, os_, self.lib.stream_of(self.q)))
        return [p[3] for p in packed], idx

    def q_forward(self, obs, w, *, nets: int = 1, target: bool = False) -> th.Tensor:
        """Q(obs_row, w) of the first ``nets`` ensemble members, eval mode: (nets, rows, A, R).  ``w``: (R,) or (rows, R)."""
        obs, w = self._f32(obs).reshape(-1, self.D), self._f32(w)
        rows = obs.shape[0]
        per_row = int(w.dim() == 2 and w.shape[0] == rows and rows > 1 or w.numel() == rows * self.R and rows > 1)
        out = th.empty((nets, rows, self.A, self.R), dtype=th.float32, device=self.q.device)
        buf = self.q_target if target else self.q
        self.lib.check_device(obs, w)
        self.lib.check(self.lib.lib.morl_gpi_q_forward(self._h, buf.data_ptr(), nets, obs.data_ptr(), w.data_ptr(), per_row,
                                                       rows, out.data_ptr(), self.lib.stream_of(self.q)))
        return out

    def action(self, obs, w, support: Optional[th.Tensor] = None):
        """(action, policy_index) device int32 scalars; ``support`` None / empty -> ``max_action``."""
        obs, w = self._f32(obs).reshape(-1), self._f32(w).reshape(-1)
        M = 0
        if support is not None and len(support) > 0:
            support = self._f32(support).reshape(-1, self.R)
            M = support.shape[0]
        res = th.zeros(2, dtype=th.int32, device=self.q.device)
        self.lib.check_device(obs, w, support if M else None)
        self.lib.check(self.lib.lib.morl_gpi_action(self._h, self.q.data_ptr(), obs.data_ptr(),
                                                    support.data_ptr() if M else None, M, w.data_ptr(), res.data_ptr(),
                                                    res[1:].data_ptr(), self.lib.stream_of(self.q)))
        return res

    def actions_batch(self, obs, w, support) -> th.Tensor:
        """GPI actions (int32, (n,)) of n observations under weight ``w`` and the support set; chunked to the workspace."""
        obs, w = self._f32(obs).reshape(-1, self.D), self._f32(w).reshape(-1)
        support = self._f32(support).reshape(-1, self.R)
        n, M = obs.shape[0], support.shape[0]
        out = th.empty(n, dtype=th.int32, device=self.q.device)
        chunk = max(1, (self.max_rows * self.max_support) // M)
        self.lib.check_device(obs, w, support)
        for b in range(0, n, chunk):
            m = min(chunk, n - b)
            self.lib.check(self.lib.lib.morl_gpi_actions(self._h, self.q.data_ptr(), obs[b:b + m].data_ptr(), m,
                                                         support.data_ptr(), M, w.data_ptr(), out[b:b + m].data_ptr(),
                                                         self.lib.stream_of(self.q)))
        return out

    def actions_rows(self, obs, w_rows, support: Optional[th.Tensor] = None) -> th.Tensor:
        """Actions (int32, (n,)) of n (observation, weight) pairs: GPI over ``support`` when given, ``max_action`` otherwise
        (``morl_gpi_actions_rows``); chunked to the workspace."""
        obs, w_rows = self._f32(obs).reshape(-1, self.D), self._f32(w_rows).reshape(-1, self.R)
        M = 0
        if support is not None and len(support) > 0:
            support = self._f32(support).reshape(-1, self.R)
            M = support.shape[0]
        n = obs.shape[0]
        if w_rows.shape[0] != n:
            raise ValueError(f"{n} observations but {w_rows.shape[0]} weight rows")
        out = th.empty(n, dtype=th.int32, device=self.q.device)
        chunk = max(1, (self.max_rows * self.max_support) // max(M, 1))
        self.lib.check_device(obs, w_rows, support if M else None)
        for b in range(0, n, chunk):
            m = min(chunk, n - b)
            self.lib.check(self.lib.lib.morl_gpi_actions_rows(
                self._h, self.q.data_ptr(), obs[b:b + m].data_ptr(), w_rows[b:b + m].data_ptr(), m,
                support.data_ptr() if M else None, M, out[b:b + m].data_ptr(), self.lib.stream_of(self.q)))
        return out

    def priority_errors(self, obs, actions, rewards, next_obs, dones, w, support=None, *, gamma=0.99, gpi_pd=True):
        obs, rewards, next_obs = self._f32(obs), self._f32(rewards), self._f32(next_obs)
        dones, w = self._f32(dones).reshape(-1), self._f32(w).reshape(-1)
        actions = th.as_tensor(actions).to(self.q.device).reshape(-1).to(th.int32).contiguous()
        rows, M = obs.shape[0], 0
        if gpi_pd:
            support = self._f32(support).reshape(-1, self.R)
            M = support.shape[0]
        out = th.empty(rows, dtype=th.float32, device=self.q.device)
        self.lib.check_device(obs, actions, rewards, next_obs, dones, w)
        self.lib.check(self.lib.lib.morl_gpi_priorities(
            self._h, self.q.data_ptr(), self.q_target.data_ptr(), obs.data_ptr(), actions.data_ptr(), rewards.data_ptr(),
            next_obs.data_ptr(), dones.data_ptr(), rows, w.data_ptr(), support.data_ptr() if M else None, M, int(gpi_pd),
            gamma, out.data_ptr(), self.lib.stream_of(self.q)))
        return out
