"""Device-resident replay buffers with the reference's API.

``ReplayBuffer`` mirrors ``common/buffer.py:20-139`` and ``PrioritizedReplayBuffer`` mirrors
``common/prioritized_buffer.py:85-226`` (same constructor arguments, ``add`` / ``sample`` / ``sample_obs`` /
``get_all_data`` / ``update_priorities`` / ``__len__``, attributes ``ptr``, ``size``, ``max_size``, ``min_priority``,
``tree``).  Differences that make the hot path fast:

* transitions are mirrored into ONE device tensor of AoS records (obs | next_obs | reward | done | action); ``add``
  stages the record in pinned host memory and the pending rows are flushed with one contiguous H2D copy per
  contiguous range right before they are needed;
* ``sample(..., to_tensor=True)`` keeps the reference's host-side index selection (same global ``np.random`` stream,
  so seeded batches are identical: ``buffer.py:82``, ``prioritized_buffer.py:40``) but gathers on the device
  (``morl_gather_batch``) and returns device tensors; only the B indices / uniforms cross PCIe;
* the PER sum tree lives on the device (float64, bit-exact arithmetic): sampling sends B uniforms, priority updates
  send nothing -- the |td . w| vector produced by the update kernel is consumed in place.

The host numpy arrays of the reference are kept as the source of truth for ``get_all_data`` / pickling.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch as th

from . import ops
from .native import NativeLib, load_library


class _HostSumTreeView:
    """Read-only host copy of the device tree with the reference's ``nodes`` layout (list of levels, root first)."""

    def __init__(self, flat: np.ndarray, n_levels: int):
        self.nodes = [flat[2 ** l - 1: 2 ** (l + 1) - 1].copy() for l in range(n_levels)]


class ReplayBuffer:
    """Multi-objective replay buffer (``common/buffer.py:20-139``) with a device mirror."""

    _PENDING = 4096  # rows staged on the host between flushes

    def __init__(self, obs_shape, action_dim, rew_dim=1, max_size=100000, obs_dtype=np.float32,
                 action_dtype=np.float32, device="cuda", lib: Optional[NativeLib] = None):
        if len(obs_shape) != 1:
            raise NotImplementedError("only vector observations are mirrored on the device")
        self.max_size = int(max_size)
        self.ptr, self.size = 0, 0
        self.obs = np.zeros((self.max_size,) + tuple(obs_shape), dtype=obs_dtype)
        self.next_obs = np.zeros((self.max_size,) + tuple(obs_shape), dtype=obs_dtype)
        self.actions = np.zeros((self.max_size, action_dim), dtype=action_dtype)
        self.rewards = np.zeros((self.max_size, rew_dim), dtype=np.float32)
        self.dones = np.zeros((self.max_size, 1), dtype=np.float32)
        self._D, self._R, self._Ad = int(obs_shape[0]), int(rew_dim), int(action_dim)
        self._int_actions = np.issubdtype(np.dtype(action_dtype), np.integer)
        self._init_device(device, lib)

    # -- device mirror ------------------------------------------------------------------------------------------
    def _init_device(self, device, lib):
        self.device = th.device(device)
        self.lib = lib or load_library()
        self._rec = 2 * self._D + self._R + 1 + self._Ad
        self.records = th.zeros((self.max_size, self._rec), dtype=th.float32, device=self.device)
        pin = self.device.type == "cuda"
        self._stage = [th.zeros((self._PENDING, self._rec), dtype=th.float32, pin_memory=pin) for _ in range(2)]
        self._stage_np = [s.numpy() for s in self._stage]
        self._stage_evt = [None, None]
        self._cur = 0
        self._pending_start = 0   # ring position of the first pending row
        self._pending_n = 0

    def _write_record(self, row: np.ndarray, obs, action, reward, next_obs, done) -> None:
        D, R = self._D, self._R
        row[:D] = np.asarray(obs, dtype=np.float32).reshape(-1)
        row[D:2 * D] = np.asarray(next_obs, dtype=np.float32).reshape(-1)
        row[2 * D:2 * D + R] = np.asarray(reward, dtype=np.float32).reshape(-1)
        row[2 * D + R] = float(done)
        row[2 * D + R + 1:] = np.asarray(action, dtype=np.float32).reshape(-1)

    def _stage_row(self, obs, action, reward, next_obs, done) -> None:
        # a flush copies at most two contiguous ring ranges, so the pending rows may wrap the ring at most once: flush as
        # soon as they would cover the whole buffer (max_size < _PENDING with learning_starts > buffer_size)
        if self._pending_n == min(self._PENDING, self.max_size):
            self.flush()
        if self._pending_n == 0:
            self._pending_start = self.ptr
            evt = self._stage_evt[self._cur]
            if evt is not None:
                evt.synchronize()     # the previous H2D out of this staging buffer must have finished
        self._write_record(self._stage_np[self._cur][self._pending_n], obs, action, reward, next_obs, done)
        self._pending_n += 1

    def flush(self) -> None:
        """Copy the pending rows to the device records (at most two contiguous ranges: the ring may wrap)."""
        n = self._pending_n
        if n == 0:
            return
        src = self._stage[self._cur]
        start = self._pending_start
        first = min(n, self.max_size - start)
        self.records[start:start + first].copy_(src[:first], non_blocking=True)
        if first < n:
            self.records[: n - first].copy_(src[first:n], non_blocking=True)
        if self.device.type == "cuda":
            evt = th.cuda.Event()
            evt.record(th.cuda.current_stream(self.device))
            self._stage_evt[self._cur] = evt
        self._on_flush(start, n)
        self._cur ^= 1
        self._pending_n = 0

    def _on_flush(self, start: int, n: int) -> None:  # PER hooks in here
        pass

    # -- reference API -------------------------------------------------------------------------------------------
    def add(self, obs, action, reward, next_obs, done):
        """``buffer.py:50-66``."""
        self._stage_row(obs, action, reward, next_obs, done)
        self.obs[self.ptr] = np.array(obs).copy()
        self.next_obs[self.ptr] = np.array(next_obs).copy()
        self.actions[self.ptr] = np.array(action).copy()
        self.rewards[self.ptr] = np.array(reward).copy()
        self.dones[self.ptr] = np.array(done).copy()
        self.ptr = (self.ptr + 1) % self.max_size
        self.size = min(self.size + 1, self.max_size)

    def add_batch(self, obs, actions, rewards, next_obs, dones):
        """``add`` for n transitions at once, in order (what the reference's Dyna loop does one call at a time,
        gpi_pd.py:394-397): host arrays and device records are written with slice copies, one H2D per contiguous range."""
        obs = np.asarray(obs, dtype=np.float32).reshape(-1, self._D)
        n = obs.shape[0]
        if n == 0:
            return
        if n > self.max_size:
            raise ValueError("batch larger than the buffer")
        self.flush()
        next_obs = np.asarray(next_obs, dtype=np.float32).reshape(n, self._D)
        rewards = np.asarray(rewards, dtype=np.float32).reshape(n, self._R)
        actions_h = np.asarray(actions).reshape(n, self._Ad)
        dones_h = np.asarray(dones, dtype=np.float32).reshape(n, 1)
        rec = np.empty((n, self._rec), dtype=np.float32)
        D, R = self._D, self._R
        rec[:, :D], rec[:, D:2 * D], rec[:, 2 * D:2 * D + R] = obs, next_obs, rewards
        rec[:, 2 * D + R] = dones_h[:, 0]
        rec[:, 2 * D + R + 1:] = actions_h.astype(np.float32)
        start = self.ptr
        first = min(n, self.max_size - start)
        rec_t = th.from_numpy(rec)
        for lo, hi, at in ((0, first, start), (first, n, 0)):
            if hi > lo:
                sl = slice(at, at + hi - lo)
                self.obs[sl], self.next_obs[sl], self.rewards[sl] = obs[lo:hi], next_obs[lo:hi], rewards[lo:hi]
                self.actions[sl], self.dones[sl] = actions_h[lo:hi].astype(self.actions.dtype), dones_h[lo:hi]
                self.records[sl].copy_(rec_t[lo:hi])
        self.ptr = (self.ptr + n) % self.max_size
        self.size = min(self.size + n, self.max_size)

    def _ring(self, name: str, count: int, dtype) -> "ops.HostRing":
        """Pinned staging ring for the host-drawn numbers of a batch: sized for the largest batch seen, a smaller batch uses
        a prefix of the slot (GPI-PD alternates between batch sizes; see ``ops.HostRing`` on why rings are not dropped)."""
        ring = ops.HostRing.fit(self.__dict__.get(name), self.lib, self.device, count, dtype)
        self.__dict__[name] = ring
        return ring

    def _gather(self, inds: np.ndarray, aux=None, prepare=None):
        """The host-drawn indices are read in place from pinned memory by the gather launch itself."""
        self.flush()
        B = int(len(inds))
        ring = self._ring("_idx_ring", B, th.int64)
        slot, ptr = ring.next(B)
        slot[:] = inds
        obs, act, rew, nobs, done, idx = ops.sample_gather(
            self.lib, self.records, B, self._D, self._R, self._Ad, self._int_actions, idx_ptr=ptr,
            aux_src_ptr=None if aux is None else aux[0], aux_dst=None if aux is None else aux[1], prepare=prepare)
        ring.mark_used()
        if self._int_actions and self._Ad == 1:
            act = act.view(-1, 1)
        return obs, act, rew, nobs, done, idx

    def draw_batches(self, batch_size: int, n: int = 1):
        """The host side of ``n`` consecutive ``sample(batch_size, to_tensor=True)`` calls for ``morl_envelope_update_n``: the
        indices of ``buffer.py:82`` drawn from the global numpy RNG in call order, staged in one pinned slot that the library's
        gather launches read in place.  Returns (address of the unit uniforms -- None here --, address of the [n][B] indices);
        call ``mark_drawn()`` once the entry that reads them has been enqueued."""
        self.flush()
        B = int(batch_size)
        ring = self._draw_ring = self._ring("_idx_ring", n * B, th.int64)
        slot, ptr = ring.next(n * B)
        for k in range(n):
            slot[k * B:(k + 1) * B] = np.random.choice(self.size, B, replace=True)
        return None, ptr

    def mark_drawn(self) -> None:
        self._draw_ring.mark_used()

    def sample(self, batch_size, replace=True, use_cer=False, to_tensor=False, device=None, aux=None, prepare=None):
        """``buffer.py:68-96``: host index selection on the global numpy RNG, device gather when ``to_tensor``.
        ``aux`` = (device-visible source address, device tensor): copied along by the gather launch (``ops.sample_gather``)."""
        inds = np.random.choice(self.size, batch_size, replace=replace)
        if use_cer:
            inds[0] = (self.ptr - 1) % self.max_size     # numpy wraps -1 to the newest slot; the device gather must too
        if to_tensor:
            return self._gather(inds, aux, prepare)
        return (self.obs[inds], self.actions[inds], self.rewards[inds], self.next_obs[inds], self.dones[inds], inds)

    def sample_obs(self, batch_size, replace=True, to_tensor=False, device=None):
        inds = np.random.choice(self.size, batch_size, replace=replace)
        if to_tensor:
            return th.tensor(self.obs[inds], device=self.device)
        return self.obs[inds]

    def get_all_data(self, max_samples=None):
        if max_samples is not None:
            inds = np.random.choice(self.size, min(max_samples, self.size), replace=False)
        else:
            inds = np.arange(self.size)
        return (self.obs[inds], self.actions[inds], self.rewards[inds], self.next_obs[inds], self.dones[inds])

    def __len__(self):
        return self.size

    # -- pickling (Envelope.save stores the buffer object, envelope.py:241-242) -----------------------------------
    def __getstate__(self):
        self.flush()
        st = {k: v for k, v in self.__dict__.items()
              if k not in ("records", "lib", "_stage", "_stage_np", "_stage_evt", "tree_dev", "running_max", "_idx_ring",
                           "_u_ring", "_draw_ring")}
        st["device"] = str(self.device)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._init_device(st["device"], None)
        self._rebuild_device()

    def _rebuild_device(self):
        n = self.size
        if n == 0:
            return
        D, R = self._D, self._R
        host = np.zeros((n, self._rec), dtype=np.float32)
        host[:, :D] = self.obs[:n]
        host[:, D:2 * D] = self.next_obs[:n]
        host[:, 2 * D:2 * D + R] = self.rewards[:n]
        host[:, 2 * D + R] = self.dones[:n, 0]
        host[:, 2 * D + R + 1:] = self.actions[:n]
        self.records[:n].copy_(th.from_numpy(host))


class PrioritizedReplayBuffer(ReplayBuffer):
    """Prioritized replay (``common/prioritized_buffer.py:85-226``) with the sum tree resident on the device."""

    TREE_BLOCK = 1024      # entries per morl_sumtree_update launch (ST_MAX_B of csrc/replay_kernels.h)

    def __init__(self, obs_shape, action_dim, rew_dim=1, max_size=100000, obs_dtype=np.float32,
                 action_dtype=np.float32, min_priority=1e-5, device="cuda", lib: Optional[NativeLib] = None):
        self._init_min_priority = float(min_priority)
        super().__init__(obs_shape, action_dim, rew_dim, max_size, obs_dtype, action_dtype, device, lib)

    def _init_device(self, device, lib):
        super()._init_device(device, lib)
        self.n_levels = int(np.ceil(np.log2(self.max_size))) + 1
        self.tree_dev = th.zeros(2 ** self.n_levels - 1, dtype=th.float64, device=self.device)
        # the reference's (mis-named) min_priority is the running MAX priority (prioritized_buffer.py:194)
        self.running_max = th.tensor([getattr(self, "_init_min_priority", 1e-5)], dtype=th.float64, device=self.device)
        self._pending_prio = []

    # the reference exposes these two as attributes
    @property
    def min_priority(self) -> float:
        return float(self.running_max.item())   # (host sync; only for callers that read it)

    @min_priority.setter
    def min_priority(self, v) -> None:
        self.running_max.fill_(float(v))

    @property
    def tree(self) -> _HostSumTreeView:
        self.flush()
        return _HostSumTreeView(self.tree_dev.cpu().numpy(), self.n_levels)

    def add(self, obs, action, reward, next_obs, done, priority=None):
        """``prioritized_buffer.py:126-147``; the tree.set is applied (in order) at the next flush."""
        self._pending_prio.append(-1.0 if priority is None else float(priority))
        super().add(obs, action, reward, next_obs, done)

    def add_batch(self, obs, actions, rewards, next_obs, dones):
        """n ``add`` calls at once: the new leaves get the running-max priority, in order (``prioritized_buffer.py:141-146``)."""
        start = self.ptr
        n = np.asarray(obs, dtype=np.float32).reshape(-1, self._D).shape[0]
        super().add_batch(obs, actions, rewards, next_obs, dones)       # (flushes pending single adds first)
        if n:
            ptr_t = th.as_tensor((start + np.arange(n)) % self.max_size, dtype=th.int64).to(self.device, non_blocking=True)
            ops.sumtree_set(self.lib, self.tree_dev, self.n_levels, ptr_t, None, self.running_max)

    def _on_flush(self, start: int, n: int) -> None:
        ptrs = (start + np.arange(n)) % self.max_size
        ptr_t = th.as_tensor(ptrs, dtype=th.int64).to(self.device, non_blocking=True)
        pr = self._pending_prio[:n]
        self._pending_prio = self._pending_prio[n:]
        val = None
        if any(p >= 0 for p in pr):
            val = th.as_tensor(pr, dtype=th.float64).to(self.device, non_blocking=True)
        ops.sumtree_set(self.lib, self.tree_dev, self.n_levels, ptr_t, val, self.running_max)

    def sample_indices(self, batch_size: int) -> th.Tensor:
        """``SumTree.sample`` (:30-54): B uniforms from the global numpy RNG (the same stream ``np.random.uniform``
        consumes), descent on the device."""
        self.flush()
        u = th.as_tensor(np.random.random_sample(batch_size)).to(self.device, non_blocking=True)
        return ops.sumtree_sample(self.lib, self.tree_dev, self.n_levels, u)

    def draw_batches(self, batch_size: int, n: int = 1):
        """See ``ReplayBuffer.draw_batches``: here the B unit uniforms of every ``SumTree.sample`` (``prioritized_buffer.py:40``,
        the stream ``np.random.uniform`` consumes), [n][B] doubles; the descents run on the device, iteration k + 1 through the
        priorities iteration k wrote.  Returns (address of the uniforms, None)."""
        self.flush()
        B = int(batch_size)
        ring = self._draw_ring = self._ring("_u_ring", n * B, th.float64)
        slot, ptr = ring.next(n * B)
        if n == 1:
            slot[:] = np.random.random_sample(B)
        else:
            for k in range(n):
                slot[k * B:(k + 1) * B] = np.random.random_sample(B)
        return ptr, None

    def sample(self, batch_size, to_tensor=False, device=None, aux=None, prepare=None):
        """``prioritized_buffer.py:149-185``.  ``to_tensor``: uniforms from the global numpy RNG (the stream
        ``np.random.uniform`` consumes) staged in pinned memory, descent + gather (+ the ``aux`` copy, see
        ``ReplayBuffer.sample``) in one launch."""
        if to_tensor:
            self.flush()
            ring = self._ring("_u_ring", int(batch_size), th.float64)
            slot, ptr = ring.next(int(batch_size))
            slot[:] = np.random.random_sample(batch_size)
            obs, act, rew, nobs, done, idx = ops.sample_gather(
                self.lib, self.records, int(batch_size), self._D, self._R, self._Ad, self._int_actions,
                tree=self.tree_dev, n_levels=self.n_levels, u01_ptr=ptr,
                aux_src_ptr=None if aux is None else aux[0], aux_dst=None if aux is None else aux[1], prepare=prepare)
            ring.mark_used()
            if self._int_actions and self._Ad == 1:
                act = act.view(-1, 1)
            return obs, act, rew, nobs, done, idx
        idx = self.sample_indices(batch_size)
        i = idx.cpu().numpy()
        return (self.obs[i], self.actions[i], self.rewards[i], self.next_obs[i], self.dones[i], i)

    def sample_obs(self, batch_size, to_tensor=False, device=None):
        i = self.sample_indices(batch_size).cpu().numpy()
        if to_tensor:
            return th.tensor(self.obs[i]).to(self.device)
        return self.obs[i]

    def update_priorities(self, idxes, priorities):
        """``prioritized_buffer.py:187-195``; float32 priorities (what every caller in the reference passes)."""
        self.flush()
        idx = th.as_tensor(idxes, dtype=th.int64).to(self.device).contiguous()
        pr = th.as_tensor(np.asarray(priorities, dtype=np.float32) if not th.is_tensor(priorities) else priorities)
        pr = pr.to(self.device, th.float32).contiguous().reshape(-1)
        if idx.numel() <= self.TREE_BLOCK:
            ops.sumtree_update(self.lib, self.tree_dev, self.n_levels, idx, pr, -1.0, self.running_max)
            return
        # more entries than one tree-update launch holds (e.g. GPIPD._reset_priorities over the whole buffer): batch_set
        # keeps, per distinct index, the priority of its FIRST occurrence and adds the leaf differences level by level in
        # ascending index order (prioritized_buffer.py:69-82) -- ascending blocks of the de-duplicated indices perform
        # exactly the same additions in the same order, so the tree stays bit-identical to the one-call result
        order = th.argsort(idx, stable=True)
        s_idx, s_pr = idx[order], pr[order]
        first = th.ones_like(s_idx, dtype=th.bool)
        first[1:] = s_idx[1:] != s_idx[:-1]
        u_idx, u_pr = s_idx[first].contiguous(), s_pr[first].contiguous()
        for b in range(0, u_idx.numel(), self.TREE_BLOCK):
            ops.sumtree_update(self.lib, self.tree_dev, self.n_levels, u_idx[b:b + self.TREE_BLOCK].contiguous(),
                               u_pr[b:b + self.TREE_BLOCK].contiguous(), -1.0, self.running_max)

    def per_update_args(self, idx: th.Tensor, alpha: float):
        """What ``ops.envelope_update(per=...)`` needs to apply ``update_priorities_from_td`` inside the gradient step."""
        self.flush()
        return (self.tree_dev, self.n_levels, idx, float(alpha), self.running_max)

    def update_priorities_from_td(self, idx: th.Tensor, raw_abs_td: th.Tensor, alpha: float) -> None:
        """Device-only path of ``envelope.py:329-334``: priority = (|td . w| + min_priority) ** alpha, then update."""
        self.flush()
        if idx.numel() > self.TREE_BLOCK:
            # more entries than one tree-update launch holds: the priorities here (every entry against the SAME running maximum, as
            # envelope.py:332 reads min_priority once), the tree through update_priorities' ascending blocks
            pr = (raw_abs_td.to(th.float32).reshape(-1) + self.running_max.to(th.float32)).pow(float(alpha))
            return self.update_priorities(idx, pr)
        ops.sumtree_update(self.lib, self.tree_dev, self.n_levels, idx, raw_abs_td, float(alpha), self.running_max)

    def get_all_data(self, max_samples=None, to_tensor=False, device=None):
        if max_samples is not None and max_samples < self.size:
            inds = np.random.choice(self.size, max_samples, replace=False)
        else:
            inds = np.arange(self.size)
        tuples = (self.obs[inds], self.actions[inds], self.rewards[inds], self.next_obs[inds], self.dones[inds])
        if to_tensor:
            return tuple(th.tensor(x).to(self.device) for x in tuples)
        return tuples

    def __getstate__(self):
        st = super().__getstate__()
        st["_tree_host"] = self.tree_dev.cpu().numpy()
        st["_running_max_host"] = float(self.running_max.item())
        st.pop("_pending_prio", None)
        return st

    def __setstate__(self, st):
        tree, rmax = st.pop("_tree_host"), st.pop("_running_max_host")
        super().__setstate__(st)
        self.tree_dev.copy_(th.from_numpy(tree))
        self.running_max.fill_(rmax)
