"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference hot path (the *oracle*).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module, and only as the checker -- the product path
(``morl-baselines_amd``) never imports anything under ``oracle/`` and has no CPU
fallback.

What is restated (reference = LucasAlegre/morl-baselines 1.3.0, paths relative to
``/root/reference/morl_baselines``):

* ``qnet_forward``            <- ``multi_policy/envelope/envelope.py:60-77`` + ``common/networks.py:10-48``
* ``envelope_target``         <- ``envelope.py:404-440`` (as written: W^2*B rows)
* ``envelope_target_dedup``   <- same result from B*W distinct rows (SURVEY.md headline fact 3)
* ``ddqn_target``             <- ``envelope.py:442-463``
* ``envelope_update``         <- ``envelope.py:267-337`` (one gradient step: targets, MSE/homotopy loss,
                                 backward, clip_grad_norm_, Adam, PER priorities)
* ``clip_grad_norm``          <- ``torch.nn.utils.clip_grad_norm_`` as used at ``envelope.py:324-325``
* ``adam_step``               <- ``torch/optim/adam.py::_single_tensor_adam`` (the CPU default), ``envelope.py:179,326``
* ``polyak_update``           <- ``common/networks.py:120-139``
* ``huber``                   <- ``common/networks.py:90-100``
* ``random_weights``          <- ``common/weights.py:10-35``
* ``linearly_decaying_value`` <- ``common/utils.py:10-32``
* ``pareto_mask`` / ``filter_pareto`` <- ``common/pareto.py:34-57`` / ``:60-73``
* ``SumTree`` / ``PrioritizedBuffer`` / ``UniformBuffer`` <- ``common/prioritized_buffer.py:12-226`` / ``common/buffer.py:20-139``

The floating-point arithmetic on this path belongs to third-party PyTorch
(``pyproject.toml:28``: ``torch >=1.12.0``, unpinned); the oracle instance is the torch
2.10.0 CPU kernels of this image.  The reference's own tests pin no numeric value on the
neural path, so the oracle is pinned instead against the reference *itself*
executed in the build container: ``tests/golden/make_golden.py`` runs the unmodified
``Envelope.update()`` through ``oracle/ref_harness.py`` and commits inputs + outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against them
(bit-exact on CPU).  Pareto parity is pinned by masks the unmodified ``get_non_pareto_dominated_inds``
produced -- on adversarial sets (``tests/golden/pareto_masks.npz``) and on the known-answer fronts of the
reference's own ``tests/test_pruning.py`` at their full sizes (``tests/golden/pruning_known_fronts.npz``,
made by ``tests/golden/make_golden_pruning.py``) -- in ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch as th
import torch.nn.functional as F

Params = List[th.Tensor]  # [W0, b0, W1, b1, ...] in nn.Linear layout: W (out, in), b (out,)


# --------------------------------------------------------------------------------------
# network
# --------------------------------------------------------------------------------------
def init_qnet_params(obs_dim: int, n_actions: int, reward_dim: int, net_arch: Sequence[int],
                     generator: Optional[th.Generator] = None) -> Params:
    """Orthogonal(gain 1) weights, zero biases -- ``networks.py:142-157`` applied by ``envelope.py:58``."""
    dims = [obs_dim + reward_dim] + list(net_arch) + [n_actions * reward_dim]
    params: Params = []
    for i in range(len(dims) - 1):
        w = th.empty(dims[i + 1], dims[i])
        th.nn.init.orthogonal_(w, gain=1, generator=generator)
        params += [w, th.zeros(dims[i + 1])]
    return params


def qnet_forward(params: Params, obs: th.Tensor, w: th.Tensor, n_actions: int, reward_dim: int) -> th.Tensor:
    """``QNet.forward`` for vector observations: cat(obs, w) -> [Linear, ReLU]*L -> Linear -> view(-1, A, R)."""
    x = th.cat((obs, w), dim=w.dim() - 1)  # envelope.py:75
    n_layers = len(params) // 2
    for l in range(n_layers):
        x = F.linear(x, params[2 * l], params[2 * l + 1])
        if l < n_layers - 1:
            x = th.relu(x)
    return x.view(-1, n_actions, reward_dim)  # envelope.py:77


# --------------------------------------------------------------------------------------
# targets
# --------------------------------------------------------------------------------------
@th.no_grad()
def envelope_target(online: Params, target: Params, obs: th.Tensor, w: th.Tensor, sampled_w: th.Tensor,
                    n_actions: int, reward_dim: int, return_indices: bool = False):
    """As written in the reference (``envelope.py:404-440``): obs is the already W-tiled next_obs (W*B rows)."""
    nW = sampled_w.size(0)
    Wrep = sampled_w.repeat(obs.size(0), 1)  # :416
    next_obs = obs.repeat_interleave(nW, 0)  # :418
    nq = qnet_forward(online, next_obs, Wrep, n_actions, reward_dim).view(obs.size(0), nW, n_actions, reward_dim)
    scal = th.einsum("br,bwar->bwa", w, nq)  # :422
    max_q, ac = th.max(scal, dim=2)  # :424
    pref = th.argmax(max_q, dim=1)  # :426
    nqt = qnet_forward(target, next_obs, Wrep, n_actions, reward_dim).view(obs.size(0), nW, n_actions, reward_dim)
    max_next_q = nqt.gather(2, ac.unsqueeze(2).unsqueeze(3).expand(nq.size(0), nq.size(1), 1, nq.size(3))).squeeze(2)
    max_next_q = max_next_q.gather(1, pref.reshape(-1, 1, 1).expand(max_next_q.size(0), 1, max_next_q.size(2))).squeeze(1)
    if return_indices:
        ac_sel = ac.gather(1, pref.reshape(-1, 1)).squeeze(1)
        return max_next_q, pref, ac_sel
    return max_next_q


@th.no_grad()
def envelope_reduce(qo: th.Tensor, qt: th.Tensor, sampled_w: th.Tensor):
    """The arg-max part of ``envelope.py:422-439`` on de-duplicated Q slabs.

    qo, qt: (B, W, A, R) with qo[b, j] = Q_online(s'_b, w_j).  Returns target (W, B, R), pref (W, B), ac (W, B)
    for TD rows r = i*B + b; the scalarisation ``w_i . Q`` uses the same batched einsum as the reference so the
    rounding (separately rounded multiply, then add, in objective order) is the reference's.
    """
    B, W, A, R = qo.shape
    # scal[i*B+b, j, a] = sum_r w_i[r] * qo[b, j, a, r], every product and every sum rounded to fp32 separately, in
    # objective order.  This is bit-identical to the reference's th.einsum("br,bwar->bwa") (envelope.py:422) whenever
    # torch evaluates that einsum with its own kernels (observed: W*A <= 132, which covers every num_sample_w the
    # reference's defaults / sweeps use); for wider slabs torch hands the contraction to the CPU BLAS, whose internal
    # accumulation order is unspecified and shape / ISA dependent (observed on this image: fma(w1,q1,w0*q0) + w2*q2
    # for R <= 3, something else for R >= 4), i.e. the reference itself is then only defined up to 1 ulp of scal.
    # The oracle therefore fixes the literal-sum rounding; tests/test_flagship_golden.py bounds the resulting
    # near-tie index differences against the reference's own output at the flagship shape.
    nq = qo.unsqueeze(0).expand(W, B, W, A, R).reshape(W * B, W, A, R)      # == next_q_values, envelope.py:420
    wr = sampled_w.repeat_interleave(B, 0)                                  # (W*B, R), row i*B+b, envelope.py:284
    scal = wr[:, None, None, 0] * nq[..., 0]
    for r in range(1, R):
        scal = scal + wr[:, None, None, r] * nq[..., r]
    max_q, ac = th.max(scal, dim=2)                                         # envelope.py:424
    pref = th.argmax(max_q, dim=1)                                          # envelope.py:426
    ac_sel = ac.gather(1, pref.unsqueeze(1)).squeeze(1)
    b_idx = th.arange(B).repeat(W)
    tgt = qt[b_idx, pref, ac_sel].view(W, B, R)
    pref, ac_sel = pref.view(W, B), ac_sel.view(W, B)
    return tgt, pref, ac_sel


@th.no_grad()
def envelope_target_dedup(online: Params, target: Params, next_obs_b: th.Tensor, sampled_w: th.Tensor,
                          n_actions: int, reward_dim: int, return_indices: bool = False):
    """Same values as ``envelope_target`` from B*W distinct rows: Q(s'_b, w_j) once per (b, j)."""
    B, nW = next_obs_b.size(0), sampled_w.size(0)
    rows_obs = next_obs_b.repeat_interleave(nW, 0)  # row (b, j)
    rows_w = sampled_w.repeat(B, 1)
    qo = qnet_forward(online, rows_obs, rows_w, n_actions, reward_dim).view(B, nW, n_actions, reward_dim)
    qt = qnet_forward(target, rows_obs, rows_w, n_actions, reward_dim).view(B, nW, n_actions, reward_dim)
    tgt, pref, ac = envelope_reduce(qo, qt, sampled_w)
    tgt = tgt.reshape(nW * B, reward_dim)
    if return_indices:
        return tgt, pref.reshape(-1), ac.reshape(-1), qo, qt
    return tgt


@th.no_grad()
def ddqn_target(online: Params, target: Params, obs: th.Tensor, w: th.Tensor, n_actions: int, reward_dim: int):
    """``envelope.py:442-463``."""
    q = qnet_forward(online, obs, w, n_actions, reward_dim)
    scal = th.einsum("br,bar->ba", w, q)
    max_acts = th.argmax(scal, dim=1)
    qt = qnet_forward(target, obs, w, n_actions, reward_dim)
    qt = qt.gather(1, max_acts.long().reshape(-1, 1, 1).expand(qt.size(0), 1, qt.size(2)))
    return qt.reshape(-1, reward_dim), max_acts


# --------------------------------------------------------------------------------------
# optimiser pieces
# --------------------------------------------------------------------------------------
@th.no_grad()
def clip_grad_norm(grads: List[th.Tensor], max_norm: float) -> th.Tensor:
    """``clip_grad_norm_`` (norm of per-tensor L2 norms; coef = max_norm/(norm+1e-6) clamped to 1; always multiplied)."""
    total = th.linalg.vector_norm(th.stack([th.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    coef = th.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


@th.no_grad()
def adam_step(params: Params, grads: List[th.Tensor], exp_avg: List[th.Tensor], exp_avg_sq: List[th.Tensor],
              step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """``_single_tensor_adam`` (no weight decay / amsgrad / capturable); ``step`` is the 1-based step being taken."""
    bias_correction1 = 1 - beta1 ** step
    bias_correction2 = 1 - beta2 ** step
    step_size = lr / bias_correction1
    bias_correction2_sqrt = bias_correction2 ** 0.5
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.lerp_(g, 1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / bias_correction2_sqrt).add_(eps)
        p.addcdiv_(m, denom, value=-step_size)


@th.no_grad()
def polyak_update(params: Params, target_params: Params, tau: float) -> None:
    """``networks.py:120-139``."""
    for p, t in zip(params, target_params):
        if tau == 1:
            t.copy_(p)
        else:
            t.mul_(1.0 - tau)
            th.add(t, p, alpha=tau, out=t)


def huber(x: th.Tensor, min_priority: float = 0.01) -> th.Tensor:
    """The reference's non-standard Huber on x=|td| (``networks.py:90-100``): no -0.5*delta^2 term."""
    return th.where(x < min_priority, 0.5 * x.pow(2), min_priority * x).mean()


# --------------------------------------------------------------------------------------
# one Envelope gradient step
# --------------------------------------------------------------------------------------
def envelope_update(online: Params, target: Params, exp_avg: List[th.Tensor], exp_avg_sq: List[th.Tensor],
                    step: int, batch: Tuple[th.Tensor, ...], sampled_w: th.Tensor, *, n_actions: int,
                    reward_dim: int, gamma: float = 0.99, lr: float = 3e-4, max_grad_norm: Optional[float] = 1.0,
                    envelope: bool = True, homotopy_lambda: float = 0.0, dedup: bool = False,
                    apply_step: bool = True) -> Dict[str, th.Tensor]:
    """One iteration of the loop body of ``Envelope.update`` (``envelope.py:269-334``) on explicit state.

    ``batch`` = (obs (B,D) f32, actions (B,1) uint8/int, rewards (B,R), next_obs (B,D), dones (B,1)) as
    ``ReplayBuffer.sample`` returns them.  ``online``/``exp_avg``/``exp_avg_sq`` are updated in place when
    ``apply_step``.  ``dedup=False`` follows the reference line by line (W^2*B-row targets); ``dedup=True``
    evaluates each distinct (b, j) row once -- tests assert both give identical bits.
    Returns every intermediate the parity tests compare.
    """
    b_obs, b_actions, b_rewards, b_next_obs, b_dones = batch
    B, nW = b_obs.size(0), sampled_w.size(0)
    out: Dict[str, th.Tensor] = {}
    w = sampled_w.repeat_interleave(B, 0)  # :284, row r = i*B + b
    t_obs = b_obs.repeat(nW, 1)
    t_act = b_actions.repeat(nW, 1)
    t_rew = b_rewards.repeat(nW, 1)
    t_nobs = b_next_obs.repeat(nW, 1)
    t_done = b_dones.repeat(nW, 1)
    with th.no_grad():
        if envelope:
            if dedup:
                tgt, pref, ac, qo, qt = envelope_target_dedup(online, target, b_next_obs, sampled_w, n_actions,
                                                              reward_dim, return_indices=True)
                out["qo"], out["qt"] = qo, qt
            else:
                tgt, pref, ac = envelope_target(online, target, t_nobs, w, sampled_w, n_actions, reward_dim,
                                                return_indices=True)
            out["pref"], out["ac"] = pref, ac
        else:
            tgt, ac = ddqn_target(online, target, t_nobs, w, n_actions, reward_dim)
            out["ac"] = ac
        target_q = t_rew + (1 - t_done) * gamma * tgt  # :298
    out["target"], out["target_q"] = tgt, target_q

    leaf = [p.detach().clone().requires_grad_(True) for p in online]
    q_values = qnet_forward(leaf, t_obs, w, n_actions, reward_dim)
    q_value = q_values.gather(1, t_act.long().reshape(-1, 1, 1).expand(q_values.size(0), 1, q_values.size(2)))
    q_value = q_value.reshape(-1, reward_dim)  # :301-305
    loss = F.mse_loss(q_value, target_q)  # :307
    if homotopy_lambda > 0:  # :309-313
        wQ = th.einsum("br,br->b", q_value, w)
        wTQ = th.einsum("br,br->b", target_q, w)
        loss = (1 - homotopy_lambda) * loss + homotopy_lambda * F.mse_loss(wQ, wTQ)
    grads = list(th.autograd.grad(loss, leaf))
    out["q_values"], out["q_value"], out["loss"] = q_values.detach(), q_value.detach(), loss.detach()
    out["grads_raw"] = [g.clone() for g in grads]
    if max_grad_norm is not None:
        out["grad_norm"] = clip_grad_norm(grads, max_grad_norm)  # :324-325
    else:
        out["grad_norm"] = th.linalg.vector_norm(th.stack([th.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    out["grads"] = grads
    if apply_step:
        adam_step(online, grads, exp_avg, exp_avg_sq, step, lr)  # :326
    # PER priority of weight-0 rows before the (p + min_priority) ** alpha host step (:330-331)
    td_err = (q_value[:B] - target_q[:B]).detach()
    out["priority_raw"] = th.einsum("sr,sr->s", td_err, w[:B]).abs()
    return out


# --------------------------------------------------------------------------------------
# host-side helpers (numpy)
# --------------------------------------------------------------------------------------
def random_weights(dim: int, n: int = 1, dist: str = "dirichlet", seed=None, rng=None) -> np.ndarray:
    """``weights.py:10-35`` (returns a 1-D vector when n == 1)."""
    if rng is None:
        rng = np.random.default_rng(seed)
    if dist == "gaussian":
        w = rng.standard_normal((n, dim))
        w = np.abs(w) / np.linalg.norm(w, ord=1, axis=1, keepdims=True)
    elif dist == "dirichlet":
        w = rng.dirichlet(np.ones(dim), n)
    else:
        raise ValueError(f"Unknown distribution {dist}")
    return w[0] if n == 1 else w


def linearly_decaying_value(initial_value, decay_period, step, warmup_steps, final_value):
    """``utils.py:10-32``."""
    steps_left = decay_period + warmup_steps - step
    bonus = (initial_value - final_value) * steps_left / decay_period
    value = final_value + bonus
    return np.clip(value, min(initial_value, final_value), max(initial_value, final_value))


def pareto_mask(candidates, remove_duplicates: bool = True) -> np.ndarray:
    """``get_non_pareto_dominated_inds`` (``pareto.py:34-57``) without the O(N^2 R) temporaries.

    keep[i] = (#{j : c_i <= c_j in every objective} == #{j : c_j == c_i}) and (first occurrence of its value
    if remove_duplicates).  The reference's second condition ``any(~res_g)`` is identically true (res_g[i,i] is
    False).  Comparisons are float64, as in the reference (evaluations are np.float64).
    """
    c = np.asarray(candidates, dtype=np.float64)
    n = c.shape[0]
    keep = np.zeros(n, dtype=bool)
    for i in range(n):
        le = np.all(c[i] <= c, axis=1)
        eq = np.all(c[i] == c, axis=1)
        ok = int(le.sum()) == int(eq.sum())
        if remove_duplicates:
            ok = ok and not bool(eq[:i].any())
        keep[i] = ok
    return keep


def filter_pareto(candidates, remove_duplicates: bool = True) -> np.ndarray:
    """``filter_pareto_dominated`` (``pareto.py:60-73``)."""
    c = np.array(candidates)
    if len(c) < 2:
        return c
    return c[pareto_mask(c, remove_duplicates)]


class SumTree:
    """``prioritized_buffer.py:12-82``: list of float64 levels, root first."""

    def __init__(self, max_size: int):
        self.nodes = []
        level_size = 1
        for _ in range(int(np.ceil(np.log2(max_size))) + 1):
            self.nodes.append(np.zeros(level_size))
            level_size *= 2

    def sample_from_uniforms(self, u01: np.ndarray) -> np.ndarray:
        """Descent for query = 0 + (total - 0) * u  (numpy's ``uniform(0, total)`` is exactly low + (high-low)*u)."""
        query = 0.0 + (self.nodes[0][0] - 0.0) * np.asarray(u01, dtype=np.float64)
        idx = np.zeros(len(query), dtype=np.int64)
        for nodes in self.nodes[1:]:
            idx *= 2
            left = nodes[idx]
            gt = np.greater(query, left)
            idx += gt
            query -= left * gt
        return idx

    def sample(self, batch_size: int) -> np.ndarray:
        return self.sample_from_uniforms(np.random.random_sample(batch_size))  # same stream as np.random.uniform

    def set(self, node_index, new_priority) -> None:
        diff = new_priority - self.nodes[-1][node_index]
        for nodes in self.nodes[::-1]:
            np.add.at(nodes, node_index, diff)
            node_index //= 2

    def batch_set(self, node_index, new_priority) -> None:
        node_index, unique_index = np.unique(node_index, return_index=True)
        diff = new_priority[unique_index] - self.nodes[-1][node_index]
        for nodes in self.nodes[::-1]:
            np.add.at(nodes, node_index, diff)
            node_index //= 2
