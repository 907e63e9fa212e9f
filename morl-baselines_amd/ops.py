"""Tensor-level wrappers over the C ABI (one function per entry point of include/morl_hip.h).

Every function takes torch tensors that live in device memory, passes raw pointers + sizes through
ctypes and enqueues work on the current torch stream; nothing here computes on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch as th

from .native import NativeLib, NetDesc, UpdateCfg, UpdateOut, _chk, _ptr, load_library, make_net_desc


DEFAULT_ENGINE = 1


class QNetContext:
    """Owns a ``morl_ctx`` (scratch workspace for a Q-network of fixed architecture)."""

    def __init__(self, obs_dim: int, reward_dim: int, n_actions: int, net_arch: Sequence[int], max_batch: int,
                 max_weights: int, lib: Optional[NativeLib] = None, fused: Optional[int] = None):
        self.lib = lib or load_library()
        self.desc: NetDesc = make_net_desc(obs_dim, reward_dim, n_actions, net_arch)
        self.obs_dim, self.reward_dim, self.n_actions = obs_dim, reward_dim, n_actions
        self.net_arch = list(net_arch)
        self.max_batch, self.max_weights = int(max_batch), int(max_weights)
        self.n_params = self.lib.param_count(self.desc)
        self.handle = self.lib.ctx_create(self.desc, self.max_batch, self.max_weights)
        # engine: 0 per-layer GEMMs, 1 layer-fused chain (row tile picked per launch), 2 / 3 row tile forced to 64 / 32;
        # default = fused whenever the architecture admits it
        self.engine = int(self.lib.lib.morl_ctx_set_fused(self.handle, DEFAULT_ENGINE if fused is None else int(fused)))
        self.fused = self.engine > 0

    def set_dw_mode(self, mode: int) -> None:
        """Weight-gradient engine: 3 balanced per-problem wave layouts with a three-stage operand pipeline (``dw_tiles.h``; default),
        2 the per-layer engine's LDS tiles as one grouped launch (the fall-back for operand rows that are not 16-byte aligned)."""
        self.lib.check(self.lib.lib.morl_ctx_set_dw_mode(self.handle, int(mode)))

    def set_timing(self, every: int) -> None:
        """Time the chain launches of every ``every``-th Envelope step with HIP events (0 / False: off, True: every step,
        -1: one launch of every step, the launches of a step taking turns)."""
        self.lib.check(self.lib.lib.morl_ctx_set_timing(self.handle, int(every)))

    def set_lazy_targets(self, enable) -> int:
        """Lazy target evaluation of ``envelope_update`` on this context (``morl_ctx_set_lazy_targets``): 0 / False never, 1 / True
        from 8 192 TD rows on (the default), 2 at every size; returns the old setting."""
        return int(self.lib.lib.morl_ctx_set_lazy_targets(self.handle, int(enable)))

    def set_exact_f32(self, enable: bool) -> bool:
        """True: every GEMM of ``envelope_update`` on the f32-input MFMA (the arithmetic of rounds 1-3); False (default): the two
        online forward passes and the dX backward pass of large steps as six split-bf16 products on the bf16 matrix cores
        (``morl_ctx_set_exact_f32``, csrc/mlp_chain_bf.h).  Returns the old setting."""
        return bool(self.lib.lib.morl_ctx_set_exact_f32(self.handle, int(bool(enable))))

    def last_step_bf16(self) -> int:
        """What the last ``envelope_update`` on this context ran as split-bf16 products: bit 0 the online passes + dX backward, bit 1
        the weight gradients; 0 = everything on the f32-input MFMA."""
        return int(self.lib.lib.morl_ctx_last_step_bf16(self.handle))

    def backpressure_seconds(self) -> float:
        """Host time this context has spent waiting for the device (``morl_ctx_backpressure_seconds``: the adaptive sizing of the lazily
        evaluated target launch reads the pair count of eight steps back)."""
        t = C.c_double(0.0)
        self.lib.check(self.lib.lib.morl_ctx_backpressure_seconds(self.handle, C.byref(t)))
        return t.value

    def lazy_target_rows(self, like: th.Tensor) -> int:
        """Distinct (transition, weight) pairs the last lazily evaluated step ran the target network on (synchronises)."""
        n = C.c_int(0)
        self.lib.check(self.lib.lib.morl_ctx_lazy_target_rows(self.handle, C.byref(n), self.lib.stream_of(like)))
        return n.value

    def invalidate_shadows(self) -> None:
        """Drop the K-major shadow copies a ``sample(prepare=...)`` launch made (``morl_ctx_invalidate_shadows``): for callers
        that write the parameter buffers in place between that launch and the step it prepared."""
        self.lib.check(self.lib.lib.morl_ctx_invalidate_shadows(self.handle))

    def read_timing(self):
        """(number of timed chain launches, their summed duration in ms); synchronises on them."""
        n, ms = C.c_int(0), C.c_double(0.0)
        self.lib.check(self.lib.lib.morl_ctx_read_timing(self.handle, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def debug_hidden(self, layer: int, rows: int, like: th.Tensor) -> th.Tensor:
        """Parity-test aid (``morl_ctx_debug_hidden``): post-ReLU activations (rows, width) of hidden layer ``layer`` saved by
        the last training forward; ``like`` fixes the device / stream."""
        out = th.empty((rows, self.desc.dims[layer]), dtype=th.float32, device=like.device)
        self.lib.check(self.lib.lib.morl_ctx_debug_hidden(self.handle, int(layer), int(rows), _ptr(out), self.lib.stream_of(like)))
        return out

    def read_timing_kinds(self):
        """{"forward" | "backward" | "dw": (launches, summed ms)} of the bracketed launches; synchronises, clears the record."""
        n, ms = (C.c_int * 4)(), (C.c_double * 4)()
        self.lib.check(self.lib.lib.morl_ctx_read_timing_kinds(self.handle, n, ms))
        return {k: (n[i], ms[i]) for i, k in enumerate(("forward", "backward", "dw", "forward2"))}

    def layer_slices(self):
        """[(w_offset, (out, in), b_offset, out)] of the flat parameter layout."""
        out, off = [], 0
        d = list(self.desc.dims)[: self.desc.n_layers + 1]
        for l in range(self.desc.n_layers):
            o, i = d[l + 1], d[l]
            out.append((off, (o, i), off + o * i, o))
            off += o * i + o
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def gather_batch(lib: NativeLib, records: th.Tensor, idx: th.Tensor, D: int, R: int, action_dim: int = 1,
                 int_actions: bool = True):
    """ReplayBuffer.sample gather (common/buffer.py:82-91) from the device record store.

    Returns (obs (B,D), actions, rewards (B,R), next_obs (B,D), dones (B,1)); actions are int32 (B,) for discrete
    action spaces (action_dim 1, int_actions) and float32 (B, action_dim) otherwise.
    """
    _chk(records, th.float32, "records")
    _chk(idx, th.int64, "idx")
    lib.check_device(records, idx)
    B = idx.numel()
    dev = records.device
    obs = th.empty((B, D), dtype=th.float32, device=dev)
    nobs = th.empty((B, D), dtype=th.float32, device=dev)
    rew = th.empty((B, R), dtype=th.float32, device=dev)
    done = th.empty((B, 1), dtype=th.float32, device=dev)
    if int_actions:
        act = th.empty((B,) if action_dim == 1 else (B, action_dim), dtype=th.int32, device=dev)
        af, ai = None, act
    else:
        act = th.empty((B, action_dim), dtype=th.float32, device=dev)
        af, ai = act, None
    lib.check(lib.lib.morl_gather_batch(_ptr(records), records.shape[1], records.shape[0], _ptr(idx), B, D, R,
                                        action_dim, _ptr(obs), _ptr(nobs), _ptr(rew), _ptr(done), _ptr(af), _ptr(ai),
                                        lib.stream_of(records)))
    return obs, act, rew, nobs, done


class HostRing:
    """A ring of small PINNED host buffers that kernels read in place (mapped into the device address space): the per-step
    host-drawn numbers (B uniforms or indices, W x R sampled weights) reach the device without a copy launch of their own.
    A slot is handed out again only after the kernel that read it has finished.  Event records cost a few microseconds of
    stream time each, so ONE event guards a whole group of slots: it is recorded after the group's last slot was used and
    waited for (normally long since complete) before the group's first slot is reused a lap later.

    Lifetime: the pinned block belongs to torch's caching HOST allocator, which knows of no stream use -- a ring that is
    dropped while a kernel still reads one of its slots could be handed out again and overwritten under that kernel.  A
    ring is therefore sized for the LARGEST count it has been asked for (``HostRing.fit``: a smaller request is a prefix of
    the slot, no re-creation when GPI-PD alternates between batch sizes) and, when it does have to be replaced, ``retire()``
    first waits for an event recorded behind its last use."""

    def __init__(self, lib: NativeLib, device: th.device, count: int, dtype, slots: int = 64, group: int = 16):
        assert slots % group == 0 and slots // group >= 2
        self.lib, self.device = lib, th.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.capacity, self.dtype = int(count), dtype
        self.buf = th.zeros((slots, count), dtype=dtype, pin_memory=self.on_gpu)
        self.np = self.buf.numpy()
        self.group = group
        self.events = [None] * (slots // group)
        self.cur = -1
        self.dev_ptrs = []
        for k in range(slots):
            host = self.buf[k].data_ptr()
            if self.on_gpu:
                out = C.c_void_p()
                lib.check(lib.lib.morl_host_device_pointer(C.c_void_p(host), C.byref(out)))
                self.dev_ptrs.append(out.value)
            else:
                self.dev_ptrs.append(host)

    @staticmethod
    def fit(ring: "Optional[HostRing]", lib: NativeLib, device, count: int, dtype) -> "HostRing":
        """``ring`` if it can hold ``count`` elements per slot, else a larger replacement (the old one retired safely)."""
        if ring is not None and ring.capacity >= count and ring.dtype == dtype:
            return ring
        if ring is not None:
            ring.retire()
            count = max(count, ring.capacity)
        return HostRing(lib, device, count, dtype)

    waited_s = 0.0        # (class-wide) host time spent blocked on the device in ``next`` -- back-pressure, not host work: a host
                          # that enqueues faster than the device executes runs into the ring a lap later (bench.py subtracts it)

    def next(self, count: Optional[int] = None):
        """(numpy view to fill -- the first ``count`` elements of the slot --, device-side address of the slot)."""
        self.cur = (self.cur + 1) % len(self.dev_ptrs)
        if self.cur % self.group == 0:
            evt = self.events[self.cur // self.group]
            if evt is not None and not evt.query():
                import time
                t0 = time.perf_counter()
                evt.synchronize()
                HostRing.waited_s += time.perf_counter() - t0
        view = self.np[self.cur]
        return (view if count is None or count == self.capacity else view[:count]), self.dev_ptrs[self.cur]

    def mark_used(self) -> None:
        """Call right after enqueueing the kernel that reads the slot handed out last."""
        if self.on_gpu and self.cur % self.group == self.group - 1:
            g = self.cur // self.group
            evt = self.events[g] or th.cuda.Event()
            evt.record(th.cuda.current_stream(self.device))
            self.events[g] = evt

    def retire(self) -> None:
        """Block until every kernel enqueued so far on the DEVICE (whatever stream read this ring's slots) has finished; only
        then may the pinned block go back to the host allocator.  Rare path (a ring outgrown)."""
        if self.on_gpu and self.cur >= 0:
            th.cuda.synchronize(self.device)


def sample_gather(lib: NativeLib, records: th.Tensor, B: int, D: int, R: int, action_dim: int = 1, int_actions: bool = True, *,
                  tree: Optional[th.Tensor] = None, n_levels: int = 0, u01_ptr: Optional[int] = None,
                  idx_ptr: Optional[int] = None, aux_src_ptr: Optional[int] = None, aux_dst: Optional[th.Tensor] = None,
                  prepare=None):
    """One training batch in one launch (``morl_sample_gather``): sum-tree descent with the uniforms at ``u01_ptr`` (PER) or
    the indices at ``idx_ptr`` (uniform replay), then the record gather; ``aux_src_ptr`` -> ``aux_dst`` rides along.  The raw
    addresses are device-visible (device memory or mapped pinned host memory of a ``HostRing``).
    ``prepare`` = (QNetContext, params_online, params_target): the launch also makes the K-major shadow weights the Envelope step
    on that context streams (``morl_envelope_prepare``), so the step itself starts with its forward passes.
    Returns (obs, actions, rewards, next_obs, dones, idx)."""
    _chk(records, th.float32, "records")
    lib.check_device(records, tree, aux_dst)
    dev = records.device
    obs = th.empty((B, D), dtype=th.float32, device=dev)
    nobs = th.empty((B, D), dtype=th.float32, device=dev)
    rew = th.empty((B, R), dtype=th.float32, device=dev)
    done = th.empty((B, 1), dtype=th.float32, device=dev)
    idx = th.empty((B,), dtype=th.int64, device=dev)
    if int_actions:
        act = th.empty((B,) if action_dim == 1 else (B, action_dim), dtype=th.int32, device=dev)
        af, ai = None, act
    else:
        act = th.empty((B, action_dim), dtype=th.float32, device=dev)
        af, ai = act, None
    tail = (_ptr(tree), int(n_levels), u01_ptr, idx_ptr, _ptr(records), records.shape[1], records.shape[0], B, D, R, action_dim,
            _ptr(obs), _ptr(nobs), _ptr(rew), _ptr(done), _ptr(af), _ptr(ai), _ptr(idx), aux_src_ptr, _ptr(aux_dst),
            0 if aux_dst is None else aux_dst.numel(), lib.stream_of(records))
    if prepare is not None:
        ctx, p_online, p_target = prepare
        _chk(p_online, th.float32, "params_online"); _chk(p_target, th.float32, "params_target")
        lib.check_device(p_online, p_target)
        lib.check(lib.lib.morl_envelope_prepare(ctx.handle, _ptr(p_online), _ptr(p_target), *tail))
    else:
        lib.check(lib.lib.morl_sample_gather(*tail))
    return obs, act, rew, nobs, done, idx


def gather_fields(lib: NativeLib, records: th.Tensor, idx: th.Tensor, fields):
    """Generic record gather: ``fields`` = [(offset, width), ...] -> one (B, width) float32 tensor per field."""
    import ctypes as C
    _chk(records, th.float32, "records")
    _chk(idx, th.int64, "idx")
    lib.check_device(records, idx)
    B, n = idx.numel(), len(fields)
    outs = [th.empty((B, w), dtype=th.float32, device=records.device) for _, w in fields]
    offs = (C.c_int32 * n)(*[int(o) for o, _ in fields])
    wids = (C.c_int32 * n)(*[int(w) for _, w in fields])
    ptrs = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    lib.check(lib.lib.morl_gather_fields(_ptr(records), records.shape[1], records.shape[0], _ptr(idx), B, n, offs, wids,
                                         ptrs, lib.stream_of(records)))
    return outs


def qnet_forward(ctx: QNetContext, params: th.Tensor, obs: th.Tensor, weights: th.Tensor, row_order: int = 0):
    """Q(obs_b, w_k) for all pairs -> (B*W, A, R); row = b*W+k (row_order 0) or k*B+b (row_order 1)."""
    lib = ctx.lib
    _chk(params, th.float32, "params"); _chk(obs, th.float32, "obs"); _chk(weights, th.float32, "weights")
    lib.check_device(params, obs, weights)
    B, W = obs.shape[0], weights.shape[0]
    q = th.empty((B * W, ctx.n_actions, ctx.reward_dim), dtype=th.float32, device=obs.device)
    lib.check(lib.lib.morl_qnet_forward(ctx.handle, _ptr(params), _ptr(obs), _ptr(weights), B, W, row_order, _ptr(q),
                                        lib.stream_of(obs)))
    return q


def qnet_forward_rows(ctx: QNetContext, params: th.Tensor, obs: th.Tensor, w: th.Tensor) -> th.Tensor:
    """Row-paired Q(obs_r, w_r) -> (rows, A, R): ``QNet.forward(obs, w)`` (envelope.py:60-77)."""
    lib = ctx.lib
    _chk(params, th.float32, "params"); _chk(obs, th.float32, "obs"); _chk(w, th.float32, "w")
    lib.check_device(params, obs, w)
    n = obs.shape[0]
    if w.shape[0] != n:
        raise ValueError("obs and w must have the same number of rows")
    q = th.empty((n, ctx.n_actions, ctx.reward_dim), dtype=th.float32, device=obs.device)
    lib.check(lib.lib.morl_qnet_forward(ctx.handle, _ptr(params), _ptr(obs), _ptr(w), n, 1, 2, _ptr(q),
                                        lib.stream_of(obs)))
    return q


def envelope_reduce_rows(lib: NativeLib, qo: th.Tensor, qt: th.Tensor, row_w: th.Tensor):
    """Envelope.envelope_target's arg-max for arbitrary rows: qo/qt (n, W, A, R), row_w (n, R)."""
    _chk(qo, th.float32, "qo"); _chk(qt, th.float32, "qt"); _chk(row_w, th.float32, "row_w")
    lib.check_device(qo, qt, row_w)
    n, W, A, R = qo.shape
    target = th.empty((n, R), dtype=th.float32, device=qo.device)
    pref = th.empty((n,), dtype=th.int32, device=qo.device)
    ac = th.empty((n,), dtype=th.int32, device=qo.device)
    lib.check(lib.lib.morl_envelope_reduce_rows(_ptr(qo), _ptr(qt), _ptr(row_w), n, W, A, R, _ptr(target), _ptr(pref),
                                                _ptr(ac), lib.stream_of(qo)))
    return target, pref, ac


def envelope_greedy_actions(ctx: QNetContext, params: th.Tensor, obs: th.Tensor, w: th.Tensor) -> th.Tensor:
    """``Envelope.max_action`` (envelope.py:389-402) for n paired (obs, w) rows in one C call -> int32 actions (n,)."""
    lib = ctx.lib
    _chk(params, th.float32, "params"); _chk(obs, th.float32, "obs"); _chk(w, th.float32, "w")
    lib.check_device(params, obs, w)
    n = obs.shape[0]
    if w.shape[0] != n:
        raise ValueError("obs and w must have the same number of rows")
    ac = th.empty((n,), dtype=th.int32, device=obs.device)
    lib.check(lib.lib.morl_envelope_greedy_actions(ctx.handle, _ptr(params), _ptr(obs), _ptr(w), n, _ptr(ac),
                                                   lib.stream_of(obs)))
    return ac


def envelope_reduce(lib: NativeLib, qo: th.Tensor, qt: th.Tensor, weights: th.Tensor, diag_only: bool = False):
    """envelope.py:422-439 on de-duplicated slabs qo/qt (B, W, A, R) -> target (W*B, R), pref, ac (W*B,)."""
    _chk(qo, th.float32, "qo"); _chk(qt, th.float32, "qt"); _chk(weights, th.float32, "weights")
    lib.check_device(qo, qt, weights)
    B, W, A, R = qo.shape
    dev = qo.device
    target = th.empty((W * B, R), dtype=th.float32, device=dev)
    pref = th.empty((W * B,), dtype=th.int32, device=dev)
    ac = th.empty((W * B,), dtype=th.int32, device=dev)
    lib.check(lib.lib.morl_envelope_reduce(_ptr(qo), _ptr(qt), _ptr(weights), B, W, A, R, int(diag_only), _ptr(target),
                                           _ptr(pref), _ptr(ac), lib.stream_of(qo)))
    return target, pref, ac


def envelope_update(ctx: QNetContext, params_online: th.Tensor, params_target: th.Tensor, grads: th.Tensor,
                    exp_avg: Optional[th.Tensor], exp_avg_sq: Optional[th.Tensor], obs: th.Tensor,
                    next_obs: th.Tensor, actions: th.Tensor, rewards: th.Tensor, dones: th.Tensor,
                    weights: th.Tensor, *, gamma: float, lr: float, adam_step: int, max_grad_norm: Optional[float],
                    homotopy_lambda: float = 0.0, envelope: bool = True, beta1: float = 0.9, beta2: float = 0.999,
                    eps: float = 1e-8, apply_step: bool = True, outputs: Optional[Dict[str, th.Tensor]] = None,
                    debug: bool = False, per=None, rows_total: int = 0) -> Dict[str, th.Tensor]:
    """One Envelope gradient step (envelope.py:269-334) entirely on the device.  ``per`` = (tree, n_levels, idx, alpha,
    running_max): the step also applies its PER priority update to the device sum tree (envelope.py:329-334), as an extra
    workgroup of the weight-gradient launch."""
    lib = ctx.lib
    for t, dt, n in ((params_online, th.float32, "params_online"), (params_target, th.float32, "params_target"),
                     (grads, th.float32, "grads"), (obs, th.float32, "obs"), (next_obs, th.float32, "next_obs"),
                     (actions, th.int32, "actions"), (rewards, th.float32, "rewards"), (dones, th.float32, "dones"),
                     (weights, th.float32, "weights")):
        _chk(t, dt, n)
    lib.check_device(params_online, params_target, grads, exp_avg, exp_avg_sq, obs, next_obs, actions, rewards, dones,
                     weights)
    B, W = obs.shape[0], weights.shape[0]
    A, R = ctx.n_actions, ctx.reward_dim
    dev = obs.device
    res = outputs if outputs is not None else {}
    if "loss" not in res:
        res["loss"] = th.empty((), dtype=th.float32, device=dev)
        res["grad_norm"] = th.empty((), dtype=th.float32, device=dev)
        res["priority"] = th.empty((B,), dtype=th.float32, device=dev)
    if debug:
        res["target"] = th.empty((W * B, R), dtype=th.float32, device=dev)
        res["pref"] = th.empty((W * B,), dtype=th.int32, device=dev)
        res["ac"] = th.empty((W * B,), dtype=th.int32, device=dev)
        res["q_online_next"] = th.empty((B, W, A, R), dtype=th.float32, device=dev)
        if debug != "lazy":          # (asking for the whole target slab makes the step evaluate it eagerly: include/morl_hip.h)
            res["q_target_next"] = th.empty((B, W, A, R), dtype=th.float32, device=dev)
        res["q_values"] = th.empty((W * B, A, R), dtype=th.float32, device=dev)
    cfg = UpdateCfg(gamma=gamma, homotopy_lambda=homotopy_lambda,
                    max_grad_norm=-1.0 if max_grad_norm is None else float(max_grad_norm), lr=lr, beta1=beta1,
                    beta2=beta2, eps=eps, adam_step=int(adam_step), envelope=int(bool(envelope)),
                    apply_step=int(bool(apply_step)))
    cfg.rows_total = int(rows_total)          # > 0: this call is one rank's share of a batch-sharded step (see distributed.py)
    if per is not None:
        tree, n_levels, idx, alpha, running_max = per
        _chk(tree, th.float64, "per tree"); _chk(idx, th.int64, "per idx"); _chk(running_max, th.float64, "per running_max")
        lib.check_device(tree, idx, running_max)
        if idx.numel() != B:
            raise ValueError("per: one sampled index per transition of the batch")
        cfg.per_tree, cfg.per_idx, cfg.per_running_max = _ptr(tree), _ptr(idx), _ptr(running_max)
        cfg.per_levels, cfg.per_alpha = int(n_levels), float(alpha)
    out = UpdateOut(**{k: _ptr(res.get(k)) for k, _ in UpdateOut._fields_})
    lib.check(lib.lib.morl_envelope_update(
        ctx.handle, _ptr(params_online), _ptr(params_target), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(obs),
        _ptr(next_obs), _ptr(actions), _ptr(rewards), _ptr(dones), _ptr(weights), B, W, C.byref(cfg), C.byref(out),
        lib.stream_of(obs)))
    return res


def _update_cfg(gamma, lr, adam_step, max_grad_norm, homotopy_lambda, envelope, beta1, beta2, eps, apply_step):
    return UpdateCfg(gamma=gamma, homotopy_lambda=homotopy_lambda,
                     max_grad_norm=-1.0 if max_grad_norm is None else float(max_grad_norm), lr=lr, beta1=beta1,
                     beta2=beta2, eps=eps, adam_step=int(adam_step), envelope=int(bool(envelope)),
                     apply_step=int(bool(apply_step)))


def envelope_slabs(ctx: QNetContext, params_online: th.Tensor, params_target: th.Tensor, next_obs: th.Tensor,
                   weights_local: th.Tensor, out: Optional[th.Tensor] = None) -> th.Tensor:
    """This rank's next-state slabs [2][B][W_local][A][R] (online, target) in one call (``morl_envelope_slabs``)."""
    lib = ctx.lib
    for t, n in ((params_online, "params_online"), (params_target, "params_target"), (next_obs, "next_obs"),
                 (weights_local, "weights_local")):
        _chk(t, th.float32, n)
    B, Wl = next_obs.shape[0], weights_local.shape[0]
    if out is None:
        out = th.empty((2, B, Wl, ctx.n_actions, ctx.reward_dim), dtype=th.float32, device=next_obs.device)
    lib.check_device(params_online, params_target, next_obs, weights_local, out)
    lib.check(lib.lib.morl_envelope_slabs(ctx.handle, _ptr(params_online), _ptr(params_target), _ptr(next_obs),
                                          _ptr(weights_local), B, Wl, _ptr(out), lib.stream_of(next_obs)))
    return out


def envelope_slab_online(ctx: QNetContext, params_online: th.Tensor, params_target: th.Tensor, next_obs: th.Tensor,
                         weights_local: th.Tensor, out: th.Tensor) -> th.Tensor:
    """This rank's ONLINE next-state slab [B][W_local][A][R] only (``morl_envelope_slab_online``: the lazily evaluated form of the
    weight-sharded step -- half the all-gather, no target pass); ``out``: at least B * W_local * A * R floats, written from its start."""
    lib = ctx.lib
    for t, n in ((params_online, "params_online"), (params_target, "params_target"), (next_obs, "next_obs"),
                 (weights_local, "weights_local"), (out, "out")):
        _chk(t, th.float32, n)
    B, Wl = next_obs.shape[0], weights_local.shape[0]
    if out.numel() < B * Wl * ctx.n_actions * ctx.reward_dim:
        raise ValueError("out is too small for the online slab")
    lib.check_device(params_online, params_target, next_obs, weights_local, out)
    lib.check(lib.lib.morl_envelope_slab_online(ctx.handle, _ptr(params_online), _ptr(params_target), _ptr(next_obs),
                                                _ptr(weights_local), B, Wl, _ptr(out), lib.stream_of(next_obs)))
    return out


def shard_lazy(ctx: QNetContext, B: int, w_local: int, envelope: bool = True) -> bool:
    """Does the one-call weight-sharded step of B x w_local rows per rank evaluate its targets lazily (``morl_ctx_shard_lazy``)?"""
    return bool(ctx.lib.lib.morl_ctx_shard_lazy(ctx.handle, int(B), int(w_local), int(bool(envelope))))


def envelope_main_forward(ctx: QNetContext, params_online: th.Tensor, obs: th.Tensor, weights_local: th.Tensor) -> None:
    """Hoisted training forward of this rank's TD rows (``morl_envelope_main_forward``): independent of the gathered
    slabs, so it can run while the all-gather is in flight."""
    lib = ctx.lib
    for t, n in ((params_online, "params_online"), (obs, "obs"), (weights_local, "weights_local")):
        _chk(t, th.float32, n)
    lib.check_device(params_online, obs, weights_local)
    lib.check(lib.lib.morl_envelope_main_forward(ctx.handle, _ptr(params_online), _ptr(obs), _ptr(weights_local),
                                                 obs.shape[0], weights_local.shape[0], lib.stream_of(obs)))


def envelope_update_shard(ctx: QNetContext, params_online: th.Tensor, grads: th.Tensor, obs: th.Tensor,
                          actions: th.Tensor, rewards: th.Tensor, dones: th.Tensor, weights_all: th.Tensor,
                          i_offset: int, w_local: int, qo_all: th.Tensor, qt_all: th.Tensor, *, gamma: float,
                          homotopy_lambda: float = 0.0, envelope: bool = True,
                          outputs: Optional[Dict[str, th.Tensor]] = None, main_forward_done: bool = False,
                          slab_parts: int = 0, lazy=None) -> Dict[str, th.Tensor]:
    """Stage B of a weight-sharded Envelope step (see include/morl_hip.h): this rank's TD rows against the gathered
    slabs; leaves the unclipped, globally normalised gradient contribution in ``grads``.  ``slab_parts`` = G > 1:
    ``qo_all`` / ``qt_all`` are views of part 0 inside the all-gathered buffer [G][2][B][W/G][A][R] (read in place)."""
    lib = ctx.lib
    for t, dt, n in ((params_online, th.float32, "params_online"), (grads, th.float32, "grads"),
                     (obs, th.float32, "obs"), (actions, th.int32, "actions"), (rewards, th.float32, "rewards"),
                     (dones, th.float32, "dones"), (weights_all, th.float32, "weights_all"),
                     (qo_all, th.float32, "qo_all"), (qt_all, th.float32, "qt_all")):
        _chk(t, dt, n)
    lib.check_device(params_online, grads, obs, actions, rewards, dones, weights_all, qo_all, qt_all)
    B, W = obs.shape[0], weights_all.shape[0]
    dev = obs.device
    res = outputs if outputs is not None else {}
    if "loss" not in res:
        res["loss"] = th.empty((), dtype=th.float32, device=dev)
        res["priority"] = th.zeros((B,), dtype=th.float32, device=dev)
    cfg = _update_cfg(gamma, 0.0, 1, None, homotopy_lambda, envelope, 0.9, 0.999, 1e-8, False)
    cfg.main_forward_done, cfg.slab_parts = int(bool(main_forward_done)), int(slab_parts)
    if lazy is not None:          # (params_target, next_obs): qo_all is the gathered ONLINE slab, the target rows are evaluated here
        p_target, next_obs = lazy
        _chk(p_target, th.float32, "params_target"); _chk(next_obs, th.float32, "next_obs")
        lib.check_device(p_target, next_obs)
        cfg.shard_params_target, cfg.shard_next_obs = _ptr(p_target), _ptr(next_obs)
    out = UpdateOut(**{k: _ptr(res.get(k)) for k, _ in UpdateOut._fields_})
    lib.check(lib.lib.morl_envelope_update_shard(
        ctx.handle, _ptr(params_online), _ptr(grads), _ptr(obs), _ptr(actions), _ptr(rewards), _ptr(dones),
        _ptr(weights_all), B, W, int(i_offset), int(w_local), _ptr(qo_all), _ptr(qt_all), C.byref(cfg), C.byref(out),
        lib.stream_of(obs)))
    return res


def envelope_step_sharded(ctx: QNetContext, comm_handle: int, params_online: th.Tensor, params_target: th.Tensor,
                          grads_x: th.Tensor, exp_avg: th.Tensor, exp_avg_sq: th.Tensor, obs: th.Tensor, next_obs: th.Tensor,
                          actions: th.Tensor, rewards: th.Tensor, dones: th.Tensor, weights_all: th.Tensor, i_offset: int,
                          w_local: int, slab_local: th.Tensor, slab_all: th.Tensor, *, gamma: float, lr: float,
                          adam_step: int, max_grad_norm: Optional[float], homotopy_lambda: float = 0.0,
                          envelope: bool = True, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                          per=None) -> None:
    """One rank's whole sharded Envelope step in one C call (``morl_envelope_step_sharded``): slabs, all-gather beside the
    training forward, TD / backward, all-reduce of ``grads_x`` = [P gradient | loss | B priorities], clip + Adam and -- with
    ``per`` = (tree, n_levels, idx, alpha, running_max) -- the PER update from the summed priorities."""
    lib = ctx.lib
    for t, dt, n in ((params_online, th.float32, "params_online"), (params_target, th.float32, "params_target"),
                     (grads_x, th.float32, "grads_x"), (exp_avg, th.float32, "exp_avg"), (exp_avg_sq, th.float32, "exp_avg_sq"),
                     (obs, th.float32, "obs"), (next_obs, th.float32, "next_obs"), (actions, th.int32, "actions"),
                     (rewards, th.float32, "rewards"), (dones, th.float32, "dones"), (weights_all, th.float32, "weights_all"),
                     (slab_local, th.float32, "slab_local"), (slab_all, th.float32, "slab_all")):
        _chk(t, dt, n)
    lib.check_device(params_online, params_target, grads_x, exp_avg, exp_avg_sq, obs, next_obs, actions, rewards, dones,
                     weights_all, slab_local, slab_all)
    B, W = obs.shape[0], weights_all.shape[0]
    P = ctx.n_params
    if grads_x.numel() != P + 1 + B:
        raise ValueError("grads_x must hold P + 1 + B floats")
    if slab_local.numel() * (W // w_local) != slab_all.numel():
        raise ValueError("slab_all must hold W / w_local parts of slab_local's size")
    cfg = _update_cfg(gamma, lr, adam_step, max_grad_norm, homotopy_lambda, envelope, beta1, beta2, eps, True)
    if per is not None:
        tree, n_levels, idx, alpha, running_max = per
        _chk(tree, th.float64, "per tree"); _chk(idx, th.int64, "per idx"); _chk(running_max, th.float64, "per running_max")
        lib.check_device(tree, idx, running_max)
        if idx.numel() != B:
            raise ValueError("per: one sampled index per transition of the batch")
        cfg.per_tree, cfg.per_idx, cfg.per_running_max = _ptr(tree), _ptr(idx), _ptr(running_max)
        cfg.per_levels, cfg.per_alpha = int(n_levels), float(alpha)
    lib.check(lib.lib.morl_envelope_step_sharded(
        ctx.handle, comm_handle, _ptr(params_online), _ptr(params_target), _ptr(grads_x), P, _ptr(exp_avg), _ptr(exp_avg_sq),
        _ptr(obs), _ptr(next_obs), _ptr(actions), _ptr(rewards), _ptr(dones), _ptr(weights_all), B, W, int(i_offset),
        int(w_local), _ptr(slab_local), _ptr(slab_all), C.byref(cfg), lib.stream_of(obs)))


def envelope_step_batch_sharded(ctx: QNetContext, comm_handle: int, params_online: th.Tensor, params_target: th.Tensor,
                                grads_x: th.Tensor, exp_avg: th.Tensor, exp_avg_sq: th.Tensor, obs: th.Tensor,
                                next_obs: th.Tensor, actions: th.Tensor, rewards: th.Tensor, dones: th.Tensor,
                                weights: th.Tensor, b_total: int, b_offset: int, *, gamma: float, lr: float, adam_step: int,
                                max_grad_norm: Optional[float], homotopy_lambda: float = 0.0, envelope: bool = True,
                                beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, per=None) -> None:
    """One rank's whole BATCH-sharded Envelope step in one C call (``morl_envelope_step_batch_sharded``): the unsharded pipeline
    on its slice of the batch, the all-reduce of ``grads_x`` = [P gradient | loss | b_total priorities], clip + Adam and -- with
    ``per`` = (tree, n_levels, idx of ALL b_total transitions, alpha, running_max) -- the PER update."""
    lib = ctx.lib
    for t, dt, n in ((params_online, th.float32, "params_online"), (params_target, th.float32, "params_target"),
                     (grads_x, th.float32, "grads_x"), (exp_avg, th.float32, "exp_avg"), (exp_avg_sq, th.float32, "exp_avg_sq"),
                     (obs, th.float32, "obs"), (next_obs, th.float32, "next_obs"), (actions, th.int32, "actions"),
                     (rewards, th.float32, "rewards"), (dones, th.float32, "dones"), (weights, th.float32, "weights")):
        _chk(t, dt, n)
    lib.check_device(params_online, params_target, grads_x, exp_avg, exp_avg_sq, obs, next_obs, actions, rewards, dones, weights)
    B, W, P = obs.shape[0], weights.shape[0], ctx.n_params
    if grads_x.numel() != P + 1 + b_total:
        raise ValueError("grads_x must hold P + 1 + b_total floats")
    cfg = _update_cfg(gamma, lr, adam_step, max_grad_norm, homotopy_lambda, envelope, beta1, beta2, eps, True)
    if per is not None:
        tree, n_levels, idx, alpha, running_max = per
        _chk(tree, th.float64, "per tree"); _chk(idx, th.int64, "per idx"); _chk(running_max, th.float64, "per running_max")
        lib.check_device(tree, idx, running_max)
        if idx.numel() != b_total:
            raise ValueError("per: one sampled index per transition of the WHOLE batch")
        cfg.per_tree, cfg.per_idx, cfg.per_running_max = _ptr(tree), _ptr(idx), _ptr(running_max)
        cfg.per_levels, cfg.per_alpha = int(n_levels), float(alpha)
    lib.check(lib.lib.morl_envelope_step_batch_sharded(
        ctx.handle, comm_handle, _ptr(params_online), _ptr(params_target), _ptr(grads_x), P, _ptr(exp_avg), _ptr(exp_avg_sq),
        _ptr(obs), _ptr(next_obs), _ptr(actions), _ptr(rewards), _ptr(dones), _ptr(weights), B, int(b_total), int(b_offset), W,
        C.byref(cfg), lib.stream_of(obs)))


def clip_adam(ctx: QNetContext, params: th.Tensor, grads: th.Tensor, exp_avg: th.Tensor, exp_avg_sq: th.Tensor, *,
              lr: float, adam_step: int, max_grad_norm: Optional[float], beta1: float = 0.9, beta2: float = 0.999,
              eps: float = 1e-8, grad_norm_out: Optional[th.Tensor] = None) -> None:
    """clip_grad_norm_ + torch's single-tensor Adam on flat buffers (envelope.py:324-326)."""
    lib = ctx.lib
    for t, n in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, th.float32, n)
    lib.check_device(params, grads, exp_avg, exp_avg_sq, grad_norm_out)
    cfg = _update_cfg(0.0, lr, adam_step, max_grad_norm, 0.0, True, beta1, beta2, eps, True)
    lib.check(lib.lib.morl_clip_adam(ctx.handle, _ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq),
                                     C.byref(cfg), _ptr(grad_norm_out), lib.stream_of(params)))


def polyak(lib: NativeLib, src: th.Tensor, dst: th.Tensor, tau: float) -> None:
    """polyak_update (common/networks.py:120-139) on flat parameter buffers."""
    _chk(src, th.float32, "src"); _chk(dst, th.float32, "dst")
    lib.check_device(src, dst)
    if src.numel() != dst.numel():
        raise ValueError("polyak: size mismatch")
    lib.check(lib.lib.morl_polyak(_ptr(src), _ptr(dst), float(tau), src.numel(), lib.stream_of(src)))


def pareto_mask(lib: NativeLib, points: th.Tensor, remove_duplicates: bool = True) -> th.Tensor:
    """get_non_pareto_dominated_inds (common/pareto.py:34-57): points (N, R) float64 -> uint8 mask (N,)."""
    _chk(points, th.float64, "points")
    lib.check_device(points)
    N, R = points.shape
    mask = th.empty((N,), dtype=th.uint8, device=points.device)
    lib.check(lib.lib.morl_pareto_mask(_ptr(points), N, R, int(remove_duplicates), _ptr(mask), lib.stream_of(points)))
    return mask


def sumtree_sample(lib: NativeLib, tree: th.Tensor, n_levels: int, u01: th.Tensor) -> th.Tensor:
    _chk(tree, th.float64, "tree"); _chk(u01, th.float64, "u01")
    lib.check_device(tree, u01)
    idx = th.empty((u01.numel(),), dtype=th.int64, device=tree.device)
    lib.check(lib.lib.morl_sumtree_sample(_ptr(tree), n_levels, _ptr(u01), u01.numel(), _ptr(idx), lib.stream_of(tree)))
    return idx


def sumtree_set(lib: NativeLib, tree: th.Tensor, n_levels: int, ptr: th.Tensor, value: Optional[th.Tensor],
                running_max: th.Tensor) -> None:
    _chk(tree, th.float64, "tree"); _chk(ptr, th.int64, "ptr"); _chk(running_max, th.float64, "running_max")
    if value is not None:
        _chk(value, th.float64, "value")
    lib.check_device(tree, ptr, value, running_max)
    lib.check(lib.lib.morl_sumtree_set(_ptr(tree), n_levels, _ptr(ptr), _ptr(value), ptr.numel(), _ptr(running_max),
                                       lib.stream_of(tree)))


def sumtree_update(lib: NativeLib, tree: th.Tensor, n_levels: int, idx: th.Tensor, raw: th.Tensor, alpha: float,
                   running_max: th.Tensor, pr_out: Optional[th.Tensor] = None) -> None:
    _chk(tree, th.float64, "tree"); _chk(idx, th.int64, "idx"); _chk(raw, th.float32, "raw")
    _chk(running_max, th.float64, "running_max")
    lib.check_device(tree, idx, raw, running_max, pr_out)
    lib.check(lib.lib.morl_sumtree_update(_ptr(tree), n_levels, _ptr(idx), _ptr(raw), idx.numel(), float(alpha),
                                          _ptr(running_max), _ptr(pr_out), lib.stream_of(tree)))
