"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (torch-CPU, autograd for the derivatives) of the continuous-action actor-critic updates that sit
under the reference's CAPQL, MOSAC (the MORL/D subproblem learner) and GPI-PD-continuous agents (SURVEY.md section 8,
rows C2/C3, M1, G7).  Every function cites the reference lines it restates.  Random draws (re-parameterisation noise,
target-policy noise, dropout masks) are INPUTS here: the reference takes them from torch's global CPU generator, which
no device kernel can replay, so the golden generator records the reference's own draws and the parity tests feed the
same numbers to the oracle and to the HIP path.

Pinned by ``tests/test_ac_oracle_golden.py`` against fixtures produced by the unmodified reference
(``tests/golden/make_golden_ac.py``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch as th
import torch.nn.functional as F

from envelope_oracle import adam_step, polyak_update

Params = List[th.Tensor]
LN_EPS = 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# mlp() of common/networks.py:10-48: [Linear, (Dropout), (LayerNorm), ReLU] x len(arch) [+ Linear(out)]
# ---------------------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class MlpSpec:
    in_dim: int
    hidden: tuple
    out_dim: int = -1          # <= 0: no output layer (a trunk)
    layer_norm: bool = False
    drop_rate: float = 0.0

    @property
    def dims(self):
        return [self.in_dim] + list(self.hidden) + ([self.out_dim] if self.out_dim > 0 else [])

    def shapes(self):
        """Parameter shapes in ``nn.Sequential.parameters()`` order."""
        out = []
        d = self.in_dim
        for h in self.hidden:
            out += [(h, d), (h,)]
            if self.layer_norm:
                out += [(h,), (h,)]
            d = h
        if self.out_dim > 0:
            out += [(self.out_dim, d), (self.out_dim,)]
        return out

    def n_params(self):
        return sum(math.prod(s) for s in self.shapes())


def init_mlp_params(spec: MlpSpec, generator: Optional[th.Generator] = None) -> Params:
    """``layer_init`` (networks.py:142-157): orthogonal gain 1 / zero bias for Linear; LayerNorm stays (1, 0)."""
    ps: Params = []
    d = spec.in_dim
    for h in spec.hidden:
        w = th.empty(h, d)
        th.nn.init.orthogonal_(w, gain=1, generator=generator)
        ps += [w, th.zeros(h)]
        if spec.layer_norm:
            ps += [th.ones(h), th.zeros(h)]
        d = h
    if spec.out_dim > 0:
        w = th.empty(spec.out_dim, d)
        th.nn.init.orthogonal_(w, gain=1, generator=generator)
        ps += [w, th.zeros(spec.out_dim)]
    return ps


def mlp_forward(spec: MlpSpec, params: Params, x: th.Tensor, drop_masks: Optional[Sequence[th.Tensor]] = None):
    """drop_masks[l]: {0,1} keep mask of hidden layer l (train mode: z * mask / (1 - p)); None -> no dropout."""
    i = 0
    for l, _ in enumerate(spec.hidden):
        x = F.linear(x, params[i], params[i + 1])
        i += 2
        if spec.drop_rate > 0.0 and drop_masks is not None:
            x = x * drop_masks[l] * (1.0 / (1.0 - spec.drop_rate))
        if spec.layer_norm:
            x = F.layer_norm(x, (x.shape[-1],), params[i], params[i + 1], LN_EPS)
            i += 2
        x = th.relu(x)
    if spec.out_dim > 0:
        x = F.linear(x, params[i], params[i + 1])
    return x


def clone(ps: Params, grad: bool = False) -> Params:
    return [p.detach().clone().requires_grad_(grad) for p in ps]


def zeros_like(ps: Params) -> Params:
    return [th.zeros_like(p) for p in ps]


# ---------------------------------------------------------------------------------------------------------------------
# CAPQL (capql.py)
# ---------------------------------------------------------------------------------------------------------------------
LOG_SIG_MAX, LOG_SIG_MIN, EPSILON = 2.0, -20.0, 1e-6


def capql_policy_sample(trunk: MlpSpec, pp: Params, obs, w, eps, scale, bias):
    """``Policy.forward`` + ``Policy.sample`` (capql.py:118-158).  pp = trunk params + [mean.W, mean.b, logstd.W,
    logstd.b]; eps = the N(0,1) draw of ``normal.rsample()``.  Returns (action, log_prob)."""
    nt = len(trunk.shapes())
    h = mlp_forward(trunk, pp[:nt], th.cat((obs, w), dim=-1))
    mean = F.linear(h, pp[nt], pp[nt + 1])
    log_std = th.clamp(F.linear(h, pp[nt + 2], pp[nt + 3]), min=LOG_SIG_MIN, max=LOG_SIG_MAX)
    std = log_std.exp()
    x_t = mean + eps * std                                   # Normal.rsample: loc + eps * scale
    y_t = th.tanh(x_t)
    action = y_t * scale + bias
    var = std ** 2                                           # Normal.log_prob
    log_prob = (-((x_t - mean) ** 2) / (2 * var) - std.log() - math.log(math.sqrt(2 * math.pi))).sum(dim=1)
    log_prob = log_prob - th.log(scale * (1 - y_t.pow(2)) + EPSILON).sum(dim=1)
    return action, log_prob.clamp(-1e3, 1e3)


def capql_policy_action(trunk: MlpSpec, pp: Params, obs, w, scale, bias):
    """``Policy.get_action`` (capql.py:127-130)."""
    nt = len(trunk.shapes())
    h = mlp_forward(trunk, pp[:nt], th.cat((obs, w), dim=-1))
    return th.tanh(F.linear(h, pp[nt], pp[nt + 1])) * scale + bias


def capql_update(qspec: MlpSpec, trunk: MlpSpec, q_nets: List[Params], tq_nets: List[Params], pol: Params,
                 q_state: Dict, p_state: Dict, batch, eps_next, eps_pi, scale, bias, *, gamma, alpha, lr, tau,
                 step) -> Dict:
    """One iteration of ``CAPQL.update`` (capql.py:321-349).  Mutates q_nets / tq_nets / pol and the Adam states
    (dicts with 'exp_avg', 'exp_avg_sq' lists over the chained parameters) in place; ``step`` is the 1-based Adam
    step of both optimisers."""
    obs, actions, w, rewards, next_obs, dones = batch
    n = len(q_nets)
    with th.no_grad():
        next_actions, log_pi = capql_policy_sample(trunk, pol, next_obs, w, eps_next, scale, bias)
        q_targets = th.stack([mlp_forward(qspec, tq, th.cat((next_obs, next_actions, w), dim=-1)) for tq in tq_nets])
        min_target_q = th.min(q_targets, dim=0)[0] - alpha * log_pi.reshape(-1, 1)
        target_q = rewards + (1 - dones.reshape(-1, 1)) * gamma * min_target_q
    qs = [clone(q, True) for q in q_nets]
    q_values = [mlp_forward(qspec, q, th.cat((obs, actions, w), dim=-1)) for q in qs]
    critic_loss = (1 / n) * sum(F.mse_loss(qv, target_q) for qv in q_values)
    flat = [p for q in qs for p in q]
    q_grads = list(th.autograd.grad(critic_loss, flat))
    with th.no_grad():
        adam_step([p for q in q_nets for p in q], q_grads, q_state["exp_avg"], q_state["exp_avg_sq"], step, lr)
    # policy update, through the UPDATED critics
    pp = clone(pol, True)
    pi, log_pi2 = capql_policy_sample(trunk, pp, obs, w, eps_pi, scale, bias)
    q_pi = th.stack([mlp_forward(qspec, q, th.cat((obs, pi, w), dim=-1)) for q in q_nets])
    min_q = (th.min(q_pi, dim=0)[0] * w).sum(dim=-1, keepdim=True)
    policy_loss = ((alpha * log_pi2) - min_q).mean()        # (B,) - (B,1) broadcasts to (B,B): capql.py:345
    p_grads = list(th.autograd.grad(policy_loss, pp))
    with th.no_grad():
        adam_step(pol, p_grads, p_state["exp_avg"], p_state["exp_avg_sq"], step, lr)
        for q, tq in zip(q_nets, tq_nets):
            polyak_update(q, tq, tau)
    return dict(critic_loss=critic_loss.detach(), policy_loss=policy_loss.detach(), q_grads=q_grads, p_grads=p_grads,
                target_q=target_q, next_actions=next_actions, log_pi_next=log_pi, pi=pi.detach(),
                log_pi=log_pi2.detach(), q_values=[qv.detach() for qv in q_values])


# ---------------------------------------------------------------------------------------------------------------------
# MOSAC (single_policy/ser/mosac_continuous_action.py) -- the learner MORL/D runs per subproblem
# ---------------------------------------------------------------------------------------------------------------------
LOG_STD_MAX, LOG_STD_MIN = 2.0, -5.0


def mosac_get_action(trunk: MlpSpec, ap: Params, obs, eps, scale, bias):
    """``MOSACActor.forward`` + ``get_action`` (mosac_continuous_action.py:98-123).  Returns (action, log_prob (B,1))."""
    nt = len(trunk.shapes())
    h = mlp_forward(trunk, ap[:nt], obs)
    mean = F.linear(h, ap[nt], ap[nt + 1])
    log_std = th.tanh(F.linear(h, ap[nt + 2], ap[nt + 3]))
    log_std = LOG_STD_MIN + 0.5 * (LOG_STD_MAX - LOG_STD_MIN) * (log_std + 1)
    std = log_std.exp()
    x_t = mean + eps * std
    y_t = th.tanh(x_t)
    action = y_t * scale + bias
    log_prob = -((x_t - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))
    log_prob = log_prob - th.log(scale * (1 - y_t.pow(2)) + 1e-6)
    return action, log_prob.sum(1, keepdim=True)


def mosac_update(qspec: MlpSpec, trunk: MlpSpec, qf: List[Params], qf_t: List[Params], actor: Params, log_alpha,
                 q_state: Dict, a_state: Dict, al_state: Dict, batch, weights, eps_next, eps_pi: Sequence,
                 eps_alpha: Sequence, scale, bias, *, gamma, tau, q_lr, policy_lr, q_step, a_step, policy_freq,
                 do_policy, do_target, autotune, alpha, target_entropy) -> Dict:
    """``MOSAC.update`` (mosac_continuous_action.py:430-489).  weights: (R,) scalarisation vector (th.matmul).
    eps_pi[k] / eps_alpha[k]: the draws of the k-th inner actor iteration.  Mutates networks / states in place;
    log_alpha is a 1-element tensor (mutated).  Returns losses, the new alpha and the gradients."""
    obs, act, rewards, next_obs, dones = batch
    alpha_t = th.tensor(float(alpha))
    with th.no_grad():
        na, nlp = mosac_get_action(trunk, actor, next_obs, eps_next, scale, bias)
        q1n = th.matmul(mlp_forward(qspec, qf_t[0], th.cat([next_obs, na], dim=-1)), weights)
        q2n = th.matmul(mlp_forward(qspec, qf_t[1], th.cat([next_obs, na], dim=-1)), weights)
        min_next = th.min(q1n, q2n) - (alpha_t * nlp).flatten()
        next_q = th.matmul(rewards, weights).flatten() + (1 - dones.flatten()) * gamma * min_next
    qs = [clone(q, True) for q in qf]
    q1 = th.matmul(mlp_forward(qspec, qs[0], th.cat([obs, act], dim=-1)), weights).flatten()
    q2 = th.matmul(mlp_forward(qspec, qs[1], th.cat([obs, act], dim=-1)), weights).flatten()
    qf1_loss, qf2_loss = F.mse_loss(q1, next_q), F.mse_loss(q2, next_q)
    q_grads = list(th.autograd.grad(qf1_loss + qf2_loss, qs[0] + qs[1]))
    with th.no_grad():
        adam_step(qf[0] + qf[1], q_grads, q_state["exp_avg"], q_state["exp_avg_sq"], q_step, q_lr)
    out = dict(qf1_loss=qf1_loss.detach(), qf2_loss=qf2_loss.detach(), q_grads=q_grads, next_q=next_q,
               actor_losses=[], alpha_losses=[], a_grads=[])
    if do_policy:
        for k in range(policy_freq):
            ap = clone(actor, True)
            pi, log_pi = mosac_get_action(trunk, ap, obs, eps_pi[k], scale, bias)
            q1p = th.matmul(mlp_forward(qspec, qf[0], th.cat([obs, pi], dim=-1)), weights)
            q2p = th.matmul(mlp_forward(qspec, qf[1], th.cat([obs, pi], dim=-1)), weights)
            min_qf_pi = th.min(q1p, q2p).view(-1)
            actor_loss = ((alpha_t * log_pi) - min_qf_pi).mean()     # (B,1) - (B,) -> (B,B): line 460
            a_grads = list(th.autograd.grad(actor_loss, ap))
            with th.no_grad():
                adam_step(actor, a_grads, a_state["exp_avg"], a_state["exp_avg_sq"], a_step + k, policy_lr)
            out["actor_losses"].append(actor_loss.detach())
            out["a_grads"].append(a_grads)
            if autotune:
                with th.no_grad():
                    _, lp = mosac_get_action(trunk, actor, obs, eps_alpha[k], scale, bias)
                la = log_alpha.detach().clone().requires_grad_(True)
                alpha_loss = (-la * (lp + target_entropy)).mean()
                g = th.autograd.grad(alpha_loss, la)[0]
                with th.no_grad():
                    adam_step([log_alpha], [g], al_state["exp_avg"], al_state["exp_avg_sq"], a_step + k, q_lr)
                alpha_t = log_alpha.detach().exp().reshape(())
                out["alpha_losses"].append(alpha_loss.detach())
    if do_target:
        with th.no_grad():
            polyak_update(qf[0], qf_t[0], tau)
            polyak_update(qf[1], qf_t[1], tau)
    out["alpha"] = float(alpha_t)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# GPI-PD / GPI-LS with continuous actions (TD3 style, gpi_pd_continuous_action.py)
# ---------------------------------------------------------------------------------------------------------------------
def td3_policy(trunk: MlpSpec, pp: Params, obs, w, scale, bias, noise=None, policy_noise=0.2, noise_clip=0.5):
    """``Policy.forward`` (gpi_pd_continuous_action.py:50-58); ``noise`` = the ``th.randn_like`` draw or None."""
    nt = len(trunk.shapes())
    h = mlp_forward(trunk, pp[:nt], th.cat((obs, w), dim=-1))
    action = th.tanh(F.linear(h, pp[nt], pp[nt + 1]))
    if noise is not None:
        n = (noise * policy_noise).clamp(-noise_clip, noise_clip)
        action = (action + n).clamp(-1, 1)
    return action * scale + bias


def gpipd_cont_update(qspec: MlpSpec, trunk: MlpSpec, q_nets: List[Params], tq_nets: List[Params], pol: Params,
                      tpol: Params, q_state: Dict, p_state: Dict, batch, w, noise, drop: Dict, scale, bias, *, gamma,
                      lr, tau, q_step, p_step, do_policy, policy_noise=0.2, noise_clip=0.5, n_per=None,
                      min_priority=0.1, per_alpha=0.6) -> Dict:
    """One iteration of ``GPIPDContinuousAction.update`` (gpi_pd_continuous_action.py:373-434).  ``batch`` is already
    doubled when the support set has > 1 element (:381-391) and ``w`` the per-row weights; ``drop`` holds the dropout
    keep masks: drop['target'][n], drop['q'][n], drop['q_pi'][n] (lists over hidden layers), or is empty."""
    obs, actions, rewards, next_obs, dones = batch
    n = len(q_nets)
    dm = lambda key, i: drop[key][i] if drop else None  # noqa: E731
    with th.no_grad():
        next_actions = td3_policy(trunk, tpol, next_obs, w, scale, bias, noise, policy_noise, noise_clip)
        q_targets = th.stack([mlp_forward(qspec, tq, th.cat((next_obs, next_actions, w), dim=-1), dm("target", i))
                              for i, tq in enumerate(tq_nets)])
        scal = th.einsum("nbr,br->nb", q_targets, w)
        inds = th.argmin(scal, dim=0, keepdim=True).reshape(1, -1, 1).expand(1, q_targets.size(1), q_targets.size(2))
        target_q = q_targets.gather(0, inds).squeeze(0)
        target_q = rewards + (1 - dones) * gamma * target_q
    qs = [clone(q, True) for q in q_nets]
    q_values = [mlp_forward(qspec, q, th.cat((obs, actions, w), dim=-1), dm("q", i)) for i, q in enumerate(qs)]
    critic_loss = (1 / n) * sum(F.mse_loss(qv, target_q) for qv in q_values)
    q_grads = list(th.autograd.grad(critic_loss, [p for q in qs for p in q]))
    with th.no_grad():
        adam_step([p for q in q_nets for p in q], q_grads, q_state["exp_avg"], q_state["exp_avg_sq"], q_step, lr)
    out = dict(critic_loss=critic_loss.detach(), q_grads=q_grads, target_q=target_q,
               q_values=[qv.detach() for qv in q_values])
    if n_per is not None:
        per = (q_values[0].detach() - target_q)[:n_per].abs() * 0.05
        per = th.einsum("br,br->b", per, w[:n_per])
        out["priority_raw"] = per
        out["priority"] = per.numpy().flatten().clip(min=min_priority) ** per_alpha
    with th.no_grad():
        for q, tq in zip(q_nets, tq_nets):
            polyak_update(q, tq, tau)
    if do_policy:
        pp = clone(pol, True)
        a = td3_policy(trunk, pp, obs, w, scale, bias)
        q_pi = (1 / n) * sum(mlp_forward(qspec, q, th.cat((obs, a, w), dim=-1), dm("q_pi", i))
                             for i, q in enumerate(q_nets))
        policy_loss = -th.einsum("br,br->b", q_pi, w).mean()
        p_grads = list(th.autograd.grad(policy_loss, pp))
        with th.no_grad():
            adam_step(pol, p_grads, p_state["exp_avg"], p_state["exp_avg_sq"], p_step, lr)
            polyak_update(pol, tpol, tau)
        out.update(policy_loss=policy_loss.detach(), p_grads=p_grads)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# MOSAC with discrete actions (single_policy/ser/mosac_discrete_action.py)
# ---------------------------------------------------------------------------------------------------------------------
def sacd_actor(pspec: MlpSpec, ap: Params, obs):
    """``MOSACDiscreteActor.get_action`` (:98-106) without the sample: (log_prob, action_probs)."""
    logits = mlp_forward(pspec, ap, obs)
    return F.log_softmax(logits, dim=1), th.softmax(logits, dim=1)


def mosac_discrete_update(qspec: MlpSpec, pspec: MlpSpec, qf: List[Params], qf_t: List[Params], actor: Params, log_alpha,
                          q_state: Dict, a_state: Dict, al_state: Dict, batch, weights, *, n_actions, reward_dim, gamma,
                          tau, q_lr, policy_lr, step, do_target, autotune, alpha, target_entropy, adam_eps=1e-4) -> Dict:
    """``MOSACDiscrete.update`` (mosac_discrete_action.py:440-503).  qspec: mlp(D -> A*R), pspec: mlp(D -> A).  batch =
    obs, actions (B,1), rewards, next_obs, dones (B,1).  Mutates networks / optimiser states / log_alpha in place."""
    obs, act, rewards, next_obs, dones = batch
    A, R = n_actions, reward_dim
    qv = lambda p, x: mlp_forward(qspec, p, x).view(-1, A, R)  # noqa: E731
    alpha_t = th.tensor(float(alpha))
    with th.no_grad():
        nlp, nprobs = sacd_actor(pspec, actor, next_obs)
        q1n = th.matmul(qv(qf_t[0], next_obs), weights)
        q2n = th.matmul(qv(qf_t[1], next_obs), weights)
        mn = (nprobs * (th.min(q1n, q2n) - alpha_t * nlp)).sum(dim=1)
        next_q = th.matmul(rewards, weights).flatten() + (1 - dones.flatten()) * gamma * mn
    qs = [clone(q, True) for q in qf]
    q1a = th.matmul(qv(qs[0], obs), weights).gather(1, act.long()).view(-1)
    q2a = th.matmul(qv(qs[1], obs), weights).gather(1, act.long()).view(-1)
    qf1_loss, qf2_loss = F.mse_loss(q1a, next_q), F.mse_loss(q2a, next_q)
    q_grads = list(th.autograd.grad(qf1_loss + qf2_loss, qs[0] + qs[1]))
    with th.no_grad():
        adam_step(qf[0] + qf[1], q_grads, q_state["exp_avg"], q_state["exp_avg_sq"], step, q_lr, eps=adam_eps)
    ap = clone(actor, True)
    log_pi, probs = sacd_actor(pspec, ap, obs)
    with th.no_grad():
        min_q = th.min(th.matmul(qv(qf[0], obs), weights), th.matmul(qv(qf[1], obs), weights))
    actor_loss = (probs * ((float(alpha) * log_pi) - min_q)).mean()
    a_grads = list(th.autograd.grad(actor_loss, ap))
    with th.no_grad():
        adam_step(actor, a_grads, a_state["exp_avg"], a_state["exp_avg_sq"], step, policy_lr, eps=adam_eps)
    out = dict(qf1_loss=qf1_loss.detach(), qf2_loss=qf2_loss.detach(), q_grads=q_grads, next_q=next_q,
               actor_loss=actor_loss.detach(), a_grads=a_grads)
    if autotune:
        la = log_alpha.detach().clone().requires_grad_(True)
        alpha_loss = (probs.detach() * (-la.exp() * (log_pi + target_entropy).detach())).mean()
        g = th.autograd.grad(alpha_loss, la)[0]
        with th.no_grad():
            adam_step([log_alpha], [g], al_state["exp_avg"], al_state["exp_avg_sq"], step, q_lr, eps=adam_eps)
        out["alpha_loss"] = alpha_loss.detach()
        alpha_t = log_alpha.detach().exp().reshape(())
    if do_target:
        with th.no_grad():
            polyak_update(qf[0], qf_t[0], tau)
            polyak_update(qf[1], qf_t[1], tau)
    out["alpha"] = float(alpha_t)
    return out
