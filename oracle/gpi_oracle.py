"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (torch-CPU, autograd for the derivatives) of GPI-PD / GPI-LS with discrete actions
(``multi_policy/gpi_pd/gpi_pd.py``).  Dropout keep-masks are inputs (the reference draws
them from torch's global generator).  Pinned by ``tests/test_gpi_oracle_golden.py`` against fixtures the unmodified
reference produced (``tests/golden/make_golden_gpi.py``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch as th
import torch.nn.functional as F

from ac_oracle import MlpSpec, clone, mlp_forward
from envelope_oracle import adam_step

Params = List[th.Tensor]


@dataclass(frozen=True)
class GpiSpec:
    D: int
    R: int
    A: int
    arch: tuple
    layer_norm: bool = True
    drop_rate: float = 0.01

    @property
    def net(self) -> MlpSpec:
        return MlpSpec(self.arch[0], tuple(self.arch[1:]), self.A * self.R, layer_norm=self.layer_norm,
                       drop_rate=self.drop_rate)

    def shapes(self):
        """``QNet.parameters()`` order (gpi_pd.py:60-69): weights_features, state_features, net."""
        h0 = self.arch[0]
        return [(h0, self.R), (h0,), (h0, self.D), (h0,)] + self.net.shapes()


def qnet_forward(spec: GpiSpec, p: Params, obs, w, drop_masks=None):
    """``QNet.forward`` (gpi_pd.py:71-76) -> (rows, A, R)."""
    wf = th.relu(F.linear(w, p[0], p[1]))
    sf = th.relu(F.linear(obs, p[2], p[3]))
    q = mlp_forward(spec.net, p[4:], sf * wf, drop_masks)
    return q.view(-1, spec.A, spec.R)


def huber(x, min_priority=0.01):
    """``common/networks.py:90-100``."""
    return th.where(x < min_priority, 0.5 * x.pow(2), min_priority * x).mean()


def envelope_target(spec: GpiSpec, tnets: List[Params], obs, w, sampled_w, drop: Optional[List] = None):
    """``GPIPD._envelope_target`` (gpi_pd.py:662-690); drop[n] = keep masks of target net n on the rows*K inputs."""
    K = sampled_w.size(0)
    W = sampled_w.unsqueeze(0).repeat(obs.size(0), 1, 1)
    next_obs = obs.unsqueeze(1).repeat(1, K, 1)
    nqt = th.stack([qnet_forward(spec, t, next_obs.reshape(-1, spec.D), W.reshape(-1, spec.R), drop[n] if drop else None)
                    .view(obs.size(0), K, spec.A, spec.R) for n, t in enumerate(tnets)])
    qv = th.einsum("br,nbpar->nbpa", w, nqt)
    mi = th.argmin(qv, dim=0).reshape(1, nqt.size(1), nqt.size(2), nqt.size(3), 1).expand(1, nqt.size(1), nqt.size(2),
                                                                                         nqt.size(3), nqt.size(4))
    nqt = nqt.gather(0, mi).squeeze(0)
    qv = th.einsum("br,bpar->bpa", w, nqt)
    max_q, ac = th.max(qv, dim=2)
    pi = th.argmax(max_q, dim=1)
    mnq = nqt.gather(2, ac.unsqueeze(2).unsqueeze(3).expand(nqt.size(0), nqt.size(1), 1, nqt.size(3))).squeeze(2)
    mnq = mnq.gather(1, pi.reshape(-1, 1, 1).expand(mnq.size(0), 1, mnq.size(2))).squeeze(1)
    return mnq, nqt


def gpi_update(spec: GpiSpec, q_nets: List[Params], tq_nets: List[Params], state: Dict, batch, w, sampled_w, drop: Dict, *,
               gamma, lr, step, min_priority, gpi_pd, max_grad_norm=None, n_per=None, alpha=0.6) -> Dict:
    """One iteration of ``GPIPD.update`` (gpi_pd.py:418-520).  batch (already doubled) = obs, actions (rows,1), rewards,
    next_obs, dones (rows,1); drop = {'target': [net][layer], 'env': ..., 'q': ...} keep masks or {}."""
    obs, actions, rewards, next_obs, dones = batch
    nn_ = len(q_nets)
    dm = lambda key, n: drop[key][n] if drop else None  # noqa: E731
    with th.no_grad():
        nq = th.stack([qnet_forward(spec, t, next_obs, w, dm("target", n)) for n, t in enumerate(tq_nets)])
        sc = th.einsum("nbar,br->nba", nq, w)
        mi = th.argmin(sc, dim=0).reshape(1, nq.size(1), nq.size(2), 1).expand(1, nq.size(1), nq.size(2), nq.size(3))
        nq = nq.gather(0, mi).squeeze(0)
        max_q = th.einsum("br,bar->ba", w, nq)
        max_acts = th.argmax(max_q, dim=1)
        qt = nq.gather(1, max_acts.long().reshape(-1, 1, 1).expand(nq.size(0), 1, nq.size(2)))
        target_q = rewards + (1 - dones) * gamma * qt.reshape(-1, spec.R)
        target_env = None
        if gpi_pd:
            te, _ = envelope_target(spec, tq_nets, next_obs, w, sampled_w, drop["env"] if drop else None)
            target_env = rewards + (1 - dones) * gamma * te
    qs = [clone(q, True) for q in q_nets]
    losses, tds, gtds = [], [], []
    for n, q in enumerate(qs):
        psi = qnet_forward(spec, q, obs, w, dm("q", n))
        psi = psi.gather(1, actions.long().reshape(-1, 1, 1).expand(psi.size(0), 1, psi.size(2))).reshape(-1, spec.R)
        td = psi - target_q
        losses.append(huber(td.abs(), min_priority=min_priority))
        tds.append(td.abs().detach())
        if gpi_pd:
            gtds.append((psi - target_env).abs().detach())
    critic_loss = (1 / nn_) * sum(losses)
    grads = list(th.autograd.grad(critic_loss, [p for q in qs for p in q]))
    norms = []
    if max_grad_norm is not None:
        npar = len(q_nets[0])
        for n in range(nn_):
            g = grads[n * npar:(n + 1) * npar]
            total = th.norm(th.stack([th.norm(x, 2.0) for x in g]), 2.0)
            coef = th.clamp(max_grad_norm / (total + 1e-6), max=1.0)
            for x in g:
                x.mul_(coef)
            norms.append(total)
    with th.no_grad():
        adam_step([p for q in q_nets for p in q], grads, state["exp_avg"], state["exp_avg_sq"], step, lr)
    out = dict(critic_loss=critic_loss.detach(), grads=grads, target_q=target_q, target_env=target_env, norms=norms)
    if n_per is not None:
        td = th.max(th.stack(tds), dim=0)[0][:n_per]
        out["td_error"] = th.einsum("br,br->b", w[:n_per], td).abs()
        out["priority"] = out["td_error"].numpy().flatten().clip(min=min_priority) ** alpha
        if gpi_pd:
            gtd = th.max(th.stack(gtds), dim=0)[0][:n_per]
            out["gtd_error"] = th.einsum("br,br->b", w[:n_per], gtd).abs()
            out["gpriority"] = out["gtd_error"].numpy().flatten().clip(min=min_priority) ** alpha
    return out


@th.no_grad()
def gpi_action(spec: GpiSpec, q0: Params, obs, w, support):
    """``GPIPD.gpi_action`` (gpi_pd.py:564-582), eval mode.  Returns (action, policy_index)."""
    M = th.stack(list(support))
    q = qnet_forward(spec, q0, obs.repeat(M.size(0), 1), M)
    sq = th.einsum("r,bar->ba", w, q)
    max_q, a = th.max(sq, dim=1)
    pi = th.argmax(max_q)
    return int(a[pi]), int(pi)


@th.no_grad()
def max_action(spec: GpiSpec, q_nets: List[Params], obs, w):
    """``GPIPD.max_action`` (gpi_pd.py:608-617)."""
    psi = th.min(th.stack([qnet_forward(spec, q, obs.reshape(1, -1), w.reshape(1, -1)) for q in q_nets]), dim=0)[0]
    return int(th.argmax(th.einsum("r,bar->ba", w, psi), dim=1))


@th.no_grad()
def reset_priority_errors(spec: GpiSpec, q_nets, tq_nets, obs, actions, rewards, next_obs, dones, w, support, *, gamma,
                          gpi_pd):
    """|w . (r + (1-d) gamma max_next - q_a)| of ``GPIPD._reset_priorities`` (gpi_pd.py:619-655), eval mode."""
    n = obs.size(0)
    qv = qnet_forward(spec, q_nets[0], obs, w.repeat(n, 1))
    q_a = qv.gather(1, actions.long().reshape(-1, 1, 1).expand(n, 1, spec.R)).squeeze(1)
    if gpi_pd:
        mnq, _ = envelope_target(spec, tq_nets, next_obs, w.repeat(n, 1), th.stack(list(support)))
    else:
        nqv = qnet_forward(spec, q_nets[0], next_obs, w.repeat(n, 1))
        ma = th.argmax(th.einsum("r,bar->ba", w, nqv), dim=1)
        qt = qnet_forward(spec, tq_nets[0], next_obs, w.repeat(n, 1))
        mnq = qt.gather(1, ma.long().reshape(-1, 1, 1).expand(n, 1, spec.R)).reshape(-1, spec.R)
    return th.einsum("r,br->b", w, (rewards + (1 - dones) * gamma * mnq - q_a)).abs()
