"""Device-resident state + one-call updates of the continuous-action actor-critic learners (CAPQL, MOSAC / MORL-D
subproblems, GPI-PD continuous) on top of ``morl_ac_*`` of the C ABI.

All parameters, target networks and Adam moments of ``population`` learners live in flat fp32 device buffers
(``q[pop][num_q][Pq]``, ``pol[pop][Pp]``); the ``nn.Parameter``s of the host-side classes are views into them, so a
whole population advances with one ``morl_ac_update`` call and no per-parameter Python loop.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch as th

from . import native
from .native import GRAD_HOOK, ACBatch, ACCfg, ACDesc, ACOut, ACState, NativeLib

ALGO_CAPQL, ALGO_MOSAC, ALGO_TD3, ALGO_SACD = 0, 1, 2, 3


class ACEngine:
    def __init__(self, algo: int, obs_dim: int, act_dim: int, reward_dim: int, net_arch: Sequence[int], *,
                 action_low, action_high, max_rows: int, num_q: int = 2, q_layer_norm: bool = False,
                 q_drop_rate: float = 0.0, population: int = 1, device="cuda", lib: Optional[NativeLib] = None,
                 device_steps: bool = False, state: Optional[Dict[str, th.Tensor]] = None):
        """``device_steps``: keep the Adam step counters of every learner on the device (needed when the learners of a
        population have taken different numbers of steps).  ``state``: adopt existing state tensors instead of
        allocating (``member()`` uses it to hand out pop-1 engines whose buffers are slices of a population's)."""
        self.lib = lib or native.load_library()
        self.device = th.device(device)
        if len(net_arch) > native.MORL_MAX_LAYERS - 1:
            raise ValueError(f"at most {native.MORL_MAX_LAYERS - 1} hidden layers are supported")
        d = ACDesc()
        d.algo, d.obs_dim, d.act_dim, d.reward_dim = algo, obs_dim, act_dim, reward_dim
        d.n_hidden = len(net_arch)
        for i, h in enumerate(net_arch):
            d.hidden[i] = int(h)
        d.num_q, d.q_layer_norm, d.q_drop_rate = num_q, int(bool(q_layer_norm)), float(q_drop_rate)
        d.population, d.max_rows = population, max_rows
        self.desc = d
        self.algo, self.D, self.Ad, self.R = algo, obs_dim, act_dim, reward_dim
        self.arch = [int(h) for h in net_arch]
        self.num_q, self.pop, self.max_rows = num_q, population, max_rows
        self.layer_norm, self.drop_rate = bool(q_layer_norm), float(q_drop_rate)
        self.heads = 1 if algo in (ALGO_TD3, ALGO_SACD) else 2
        self.w_input = algo not in (ALGO_MOSAC, ALGO_SACD)
        self.discrete = algo == ALGO_SACD            # act_dim = number of actions; critics output A * R values
        self.Pq = int(self.lib.lib.morl_ac_q_param_count(C.byref(d)))
        self.Pp = int(self.lib.lib.morl_ac_policy_param_count(C.byref(d)))
        if self.Pq < 0 or self.Pp < 0:
            self.lib.check(-1)
        h = C.c_void_p()
        self.lib.check(self.lib.lib.morl_ac_create(C.byref(h), C.byref(d)))
        self._h = h.value
        self._ctor = dict(algo=algo, obs_dim=obs_dim, act_dim=act_dim, reward_dim=reward_dim, net_arch=list(net_arch),
                          action_low=action_low, action_high=action_high, max_rows=max_rows, num_q=num_q,
                          q_layer_norm=q_layer_norm, q_drop_rate=q_drop_rate, device=device, lib=self.lib)
        if state is not None:
            for name in native.AC_STATE_FIELDS:
                setattr(self, name, state.get(name))
        else:
            z = lambda *shape: th.zeros(*shape, dtype=th.float32, device=self.device)  # noqa: E731
            self.q, self.q_target = z(population, num_q, self.Pq), z(population, num_q, self.Pq)
            self.q_exp_avg, self.q_exp_avg_sq = z(population, num_q, self.Pq), z(population, num_q, self.Pq)
            self.pol, self.pol_exp_avg, self.pol_exp_avg_sq = (z(population, self.Pp) for _ in range(3))
            self.pol_target = z(population, self.Pp) if algo == ALGO_TD3 else None
            self.log_alpha = self.log_alpha_exp_avg = self.log_alpha_exp_avg_sq = None
            if algo in (ALGO_MOSAC, ALGO_SACD):
                self.log_alpha, self.log_alpha_exp_avg, self.log_alpha_exp_avg_sq = (z(population) for _ in range(3))
            low = np.broadcast_to(np.asarray(action_low, dtype=np.float32), (act_dim,))
            high = np.broadcast_to(np.asarray(action_high, dtype=np.float32), (act_dim,))
            self.action_scale = th.tensor((high - low) / 2.0, dtype=th.float32, device=self.device)
            self.action_bias = th.tensor((high + low) / 2.0, dtype=th.float32, device=self.device)
            self.q_steps = self.pol_steps = None
            if device_steps:
                self.q_steps = th.zeros(population, dtype=th.int32, device=self.device)
                self.pol_steps = th.zeros(population, dtype=th.int32, device=self.device)
        self.lib.check_device(self.q)

    def member(self, k: int, max_rows: Optional[int] = None) -> "ACEngine":
        """A population-1 engine (own workspace) whose state tensors are learner ``k``'s slices of this engine's."""
        state = {}
        for name in native.AC_STATE_FIELDS:
            t = getattr(self, name)
            state[name] = t if (t is None or name in ("action_scale", "action_bias")) else t[k:k + 1]
        kw = dict(self._ctor)
        if max_rows is not None:
            kw["max_rows"] = max_rows
        algo, D, Ad, R, arch = (kw.pop(k_) for k_ in ("algo", "obs_dim", "act_dim", "reward_dim", "net_arch"))
        return ACEngine(algo, D, Ad, R, arch, population=1, state=state, **kw)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.lib.morl_ac_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- parameter views (torch nn.Sequential.parameters() order) ---------------------------------------------------
    def _q_shapes(self):
        out, d = [], (self.D if self.discrete else self.D + self.Ad + (self.R if self.w_input else 0))
        for hdim in self.arch:
            out += [(hdim, d), (hdim,)]
            if self.layer_norm:
                out += [(hdim,), (hdim,)]
            d = hdim
        n_out = self.Ad * self.R if self.discrete else self.R
        return out + [(n_out, d), (n_out,)]

    @staticmethod
    def _views(flat: th.Tensor, shapes) -> List[th.Tensor]:
        out, o = [], 0
        for s in shapes:
            n = int(np.prod(s))
            out.append(flat[o:o + n].view(s))
            o += n
        assert o == flat.numel(), (o, flat.numel())
        return out

    def q_views(self, buf: th.Tensor, p: int = 0, n: int = 0) -> List[th.Tensor]:
        """Views of critic ``n`` of learner ``p`` inside ``buf`` (self.q, self.q_target, an Adam moment buffer)."""
        return self._views(buf[p, n], self._q_shapes())

    def policy_views(self, buf: th.Tensor, p: int = 0) -> List[th.Tensor]:
        """Trunk W/b pairs, then the heads in the reference's order: mean.W, mean.b[, log_std.W, log_std.b]."""
        shapes, d = [], self.D + (self.R if self.w_input else 0)
        for hdim in self.arch:
            shapes += [(hdim, d), (hdim,)]
            d = hdim
        shapes += [(self.heads * self.Ad, d), (self.heads * self.Ad,)]
        v = self._views(buf[p], shapes)
        hw, hb = v[-2], v[-1]
        out = v[:-2] + [hw[:self.Ad], hb[:self.Ad]]
        if self.heads == 2:
            out += [hw[self.Ad:], hb[self.Ad:]]
        return out

    # -- calls ----------------------------------------------------------------------------------------------------------
    def _state(self, first: int = 0) -> ACState:
        st = ACState()
        for name in native.AC_STATE_FIELDS:
            t = getattr(self, name)
            if t is None:
                setattr(st, name, None)
            elif name in ("action_scale", "action_bias"):
                setattr(st, name, t.data_ptr())
            else:
                setattr(st, name, t[first:].data_ptr() if first else t.data_ptr())
        return st

    def _f32(self, t, name) -> th.Tensor:
        t = th.as_tensor(t)
        if t.dtype != th.float32 or t.device != self.q.device or not t.is_contiguous():
            t = t.to(self.q.device, th.float32).contiguous()
        return t

    def make_cfg(self, *, gamma=0.99, tau=0.005, alpha=0.2, q_lr=3e-4, policy_lr=3e-4, alpha_lr=None, q_step=1,
                 policy_step=1, do_policy=True, policy_iters=1, do_target=True, autotune=False, target_entropy=0.0,
                 policy_noise=0.2, noise_clip=0.5, n_per=0, dropout_seed=0, beta1=0.9, beta2=0.999, eps=1e-8) -> ACCfg:
        c = ACCfg()
        c.gamma, c.tau, c.alpha = gamma, tau, alpha
        c.q_lr, c.policy_lr, c.alpha_lr = q_lr, policy_lr, (q_lr if alpha_lr is None else alpha_lr)
        c.beta1, c.beta2, c.eps = beta1, beta2, eps
        c.q_step, c.policy_step, c.do_policy, c.policy_iters = q_step, policy_step, int(do_policy), policy_iters
        c.do_target, c.autotune, c.target_entropy = int(do_target), int(autotune), target_entropy
        c.policy_noise, c.noise_clip, c.n_per, c.dropout_seed = policy_noise, noise_clip, n_per, dropout_seed
        return c

    def update(self, cfg: ACCfg, *, obs, actions, rewards, next_obs, dones, w, eps_next=None, eps_pi=None, eps_alpha=None,
               drop_masks: Optional[th.Tensor] = None, want: Sequence[str] = ("critic_loss", "policy_loss"),
               first: int = 0, count: Optional[int] = None, grad_sync=None) -> Dict:
        """One ``morl_ac_update``.  Array shapes as in include/morl_hip.h (leading [pop] axis may be omitted when
        population == 1).  ``first`` / ``count``: advance only learners first .. first+count-1 (the arrays then hold
        ``count`` learners).  Returns the requested device outputs (no host synchronisation).
        ``grad_sync(which, grads)``: data-parallel job -- reduce the critic (0) / actor (1) gradient tensor in place over
        the processes before its Adam step (``morl_ac_cfg.grad_hook``; see ``distributed.average_gradients``)."""
        obs = self._f32(obs, "obs")
        full_pop = self.pop
        count = full_pop - first if count is None else count
        if first < 0 or count < 1 or first + count > full_pop:
            raise ValueError(f"learners {first}..{first + count - 1} outside the population of {full_pop}")
        return self._update(cfg, obs, count, first, actions, rewards, next_obs, dones, w, eps_next, eps_pi, eps_alpha,
                            drop_masks, want, grad_sync)

    def _pack(self, cfg, obs, pop, actions, rewards, next_obs, dones, w, eps_next, eps_pi, eps_alpha, drop_masks, want):
        """The C structs of one ``morl_ac_update``: (ACBatch, ACOut, {name: output tensor}, tensors to keep alive)."""
        rows = obs.numel() // (pop * self.D)
        keep = [obs]
        b = ACBatch()
        b.rows = rows
        b.active = pop
        b.obs = obs.data_ptr()
        for name, t in (("actions", actions), ("rewards", rewards), ("next_obs", next_obs), ("dones", dones), ("w", w),
                        ("eps_next", eps_next), ("eps_pi", eps_pi), ("eps_alpha", eps_alpha)):
            if t is None:
                continue
            t = self._f32(t, name)
            keep.append(t)
            setattr(b, name, t.data_ptr())
        expect = dict(actions=pop * rows * (1 if self.discrete else self.Ad), rewards=pop * rows * self.R,
                      next_obs=obs.numel(), dones=pop * rows, w=pop * (rows if self.w_input else 1) * self.R)
        if not self.discrete:
            expect["eps_next"] = pop * rows * self.Ad
        for (name, n), t in zip(expect.items(), keep[1:1 + len(expect)]):
            if t.numel() != n:
                raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")
        if drop_masks is not None:
            need = int(self.lib.lib.morl_ac_mask_bytes(C.byref(self.desc), rows)) // self.pop * pop
            if drop_masks.dtype != th.uint8 or drop_masks.numel() != need or not drop_masks.is_contiguous():
                raise ValueError(f"drop_masks: expected {need} contiguous uint8 flags")
            keep.append(drop_masks)
            b.drop_masks = drop_masks.data_ptr()
        self.lib.check_device(*keep)
        o, res = ACOut(), {}
        shapes = dict(critic_loss=(pop,), q_losses=(pop, self.num_q), policy_loss=(pop,), alpha_loss=(pop,),
                      alpha=(pop,), priority=(pop, max(cfg.n_per, 1)),
                      target_q=(pop, rows) if self.algo in (ALGO_MOSAC, ALGO_SACD) else (pop, rows, self.R),
                      q_grads=(pop, self.num_q, self.Pq), pol_grads=(pop, self.Pp))
        for name in want:
            res[name] = th.zeros(shapes[name], dtype=th.float32, device=self.q.device)
            setattr(o, name, res[name].data_ptr())
        return b, o, res, keep

    def _update(self, cfg, obs, pop, first, actions, rewards, next_obs, dones, w, eps_next, eps_pi, eps_alpha,
                drop_masks, want, grad_sync=None):
        hook_error: List[BaseException] = []
        if grad_sync is not None:
            want = tuple(want) + tuple(n for n in ("q_grads",) + (("pol_grads",) if cfg.do_policy else ()) if n not in want)
        b, o, res, _keep = self._pack(cfg, obs, pop, actions, rewards, next_obs, dones, w, eps_next, eps_pi, eps_alpha,
                                      drop_masks, want)
        hook = None
        if grad_sync is not None:
            def _hook(_user, which, _ptr, _count, _stream):
                try:
                    grad_sync(int(which), res["q_grads" if which == 0 else "pol_grads"])
                    return 0
                except BaseException as e:  # noqa: BLE001  (must not propagate through the C frame)
                    hook_error.append(e)
                    return 1
            hook = GRAD_HOOK(_hook)
            cfg.grad_hook = C.cast(hook, C.c_void_p).value
            cfg.grad_hook_user = None
        st = self._state(first)
        try:
            rc = self.lib.lib.morl_ac_update(self._h, C.byref(st), C.byref(b), C.byref(cfg), C.byref(o),
                                             self.lib.stream_of(self.q))
        finally:
            cfg.grad_hook = None
        if hook_error:
            raise hook_error[0]
        self.lib.check(rc)
        return res

    def update_n(self, items: Sequence[dict], want: Sequence[str] = ("critic_loss", "policy_loss")) -> List[Dict]:
        """``morl_ac_update_n``: the reference's ``for _ in range(self.gradient_updates)`` loop in ONE library entry.  ``items``:
        for each of the n updates a dict with ``cfg`` (its own ``ACCfg``: Adam step indices, dropout seed, do_policy ...) and the
        batch keywords of ``update`` (already drawn: the host consumes its RNG streams in the reference's order beforehand).
        Every learner of the population takes part; no ``grad_sync``.  Returns the n output dicts."""
        n = len(items)
        bs, cs, os_ = (ACBatch * n)(), (ACCfg * n)(), (ACOut * n)()
        results, keep = [], []
        for k, it in enumerate(items):
            it = dict(it)
            cfg = it.pop("cfg")
            obs = self._f32(it.pop("obs"), "obs")
            b, o, res, kp = self._pack(cfg, obs, self.pop, it.get("actions"), it.get("rewards"), it.get("next_obs"), it.get("dones"),
                                       it.get("w"), it.get("eps_next"), it.get("eps_pi"), it.get("eps_alpha"), it.get("drop_masks"),
                                       it.get("want", want))
            bs[k], cs[k], os_[k] = b, cfg, o
            results.append(res)
            keep.append(kp)
        st = self._state(0)
        self.lib.check(self.lib.lib.morl_ac_update_n(self._h, C.byref(st), n, bs, cs, os_, self.lib.stream_of(self.q)))
        return results

    def update_n_per(self, items: Sequence[dict], *, buffer, u01: np.ndarray, doubled: bool, alpha: float, min_priority: float,
                     want: Sequence[str] = ("critic_loss", "policy_loss", "priority")):
        """``morl_ac_update_n_per``: the loop with prioritised replay in ONE library entry (one learner): iteration k samples
        ``buffer`` through its sum tree with the unit uniforms ``u01[k]``, gathers the transitions into the batch tensors of
        ``items[k]`` (scratch the caller allocated; w / noise filled), updates, and writes max(priority, min_priority) ** alpha
        back into the tree.  Returns (output dicts, sampled indices [n][B])."""
        from .native import GPIPer
        n = len(items)
        u01 = np.ascontiguousarray(u01, dtype=np.float64).reshape(n, -1)
        B = u01.shape[1]
        # the library gathers records straight into the batch tensors: the buffer must live where the engine does and hold what
        # this engine's batches hold (a mismatch would be out-of-bounds gather writes, not an exception)
        self.lib.check_device(buffer.tree_dev, buffer.running_max, buffer.records)
        if buffer._Ad != self.Ad or buffer._int_actions or buffer._D != self.D or buffer._R != self.R:
            raise ValueError(f"update_n_per: the buffer holds (obs {buffer._D}, reward {buffer._R}, "
                             f"{'int' if buffer._int_actions else 'float'} actions {buffer._Ad}), the engine "
                             f"(obs {self.D}, reward {self.R}, float actions {self.Ad})")
        copies = 2 if doubled else 1
        buffer.flush()
        bs, cs, os_ = (ACBatch * n)(), (ACCfg * n)(), (ACOut * n)()
        results, keep = [], []
        for k, it in enumerate(items):
            it = dict(it)
            cfg = it.pop("cfg")
            obs = self._f32(it.pop("obs"), "obs")
            if obs.numel() != copies * B * self.D:
                raise ValueError(f"update_n_per: update {k} has {obs.numel() // self.D} rows, {copies} x {B} sampled transitions expected")
            b, o, res, kp = self._pack(cfg, obs, self.pop, it.get("actions"), it.get("rewards"), it.get("next_obs"), it.get("dones"),
                                       it.get("w"), it.get("eps_next"), it.get("eps_pi"), it.get("eps_alpha"), it.get("drop_masks"),
                                       it.get("want", want))
            bs[k], cs[k], os_[k] = b, cfg, o
            results.append(res)
            keep.append(kp)
        dev = self.q.device
        u_dev = th.as_tensor(u01).to(dev)
        idx = th.empty((n, B), dtype=th.int64, device=dev)
        per = GPIPer()
        per.tree, per.running_max, per.u01 = buffer.tree_dev.data_ptr(), buffer.running_max.data_ptr(), u_dev.data_ptr()
        per.records, per.idx = buffer.records.data_ptr(), idx.data_ptr()
        per.capacity, per.record_floats = buffer.records.shape[0], buffer.records.shape[1]
        per.n_levels, per.D, per.R, per.action_dim, per.B = buffer.n_levels, buffer._D, buffer._R, buffer._Ad, B
        per.doubled, per.use_gtd, per.alpha, per.min_priority = int(doubled), 0, float(alpha), float(min_priority)
        st = self._state(0)
        self.lib.check(self.lib.lib.morl_ac_update_n_per(self._h, C.byref(st), n, C.byref(per), bs, cs, os_,
                                                         self.lib.stream_of(self.q)))
        return results, idx

    def policy_forward(self, obs, w=None, *, eps=None, use_target=False, cfg: Optional[ACCfg] = None,
                       want_logp=False):
        obs = self._f32(obs, "obs")
        rows = obs.numel() // (self.pop * self.D)
        w = None if w is None else self._f32(w, "w")
        eps = None if eps is None else self._f32(eps, "eps")
        act = th.empty((self.pop, rows, self.Ad), dtype=th.float32, device=self.q.device)
        logp = th.empty((self.pop, rows), dtype=th.float32, device=self.q.device) if want_logp else None
        cfg = cfg or self.make_cfg()
        st = self._state()
        self.lib.check_device(obs, w, eps)
        self.lib.check(self.lib.lib.morl_ac_policy_forward(
            self._h, C.byref(st), obs.data_ptr(), None if w is None else w.data_ptr(), rows, 0 if eps is None else 1,
            None if eps is None else eps.data_ptr(), int(use_target), C.byref(cfg), act.data_ptr(),
            None if logp is None else logp.data_ptr(), self.lib.stream_of(self.q)))
        return (act, logp) if want_logp else act

    def q_forward(self, obs, actions=None, w=None, *, use_target=False) -> th.Tensor:
        obs = self._f32(obs, "obs")
        actions = None if actions is None else self._f32(actions, "actions")
        rows = obs.numel() // (self.pop * self.D)
        w = None if w is None else self._f32(w, "w")
        n_out = self.Ad * self.R if self.discrete else self.R
        out = th.empty((self.pop, self.num_q, rows, n_out), dtype=th.float32, device=self.q.device)
        st = self._state()
        self.lib.check_device(obs, actions, w)
        self.lib.check(self.lib.lib.morl_ac_q_forward(
            self._h, C.byref(st), obs.data_ptr(), None if actions is None else actions.data_ptr(),
            None if w is None else w.data_ptr(), rows,
            int(use_target), out.data_ptr(), self.lib.stream_of(self.q)))
        return out
