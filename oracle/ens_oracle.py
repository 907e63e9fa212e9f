"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement of the probabilistic dynamics ensemble (``common/model_based/probabilistic_ensemble.py``): forward with
the bounded log-variance, ``_compute_loss``, the optimiser step ``fit`` takes (Adam with per-layer weight decay) and
``_compute_mse_losses``.  Parameters in the reference's layout: ``W[l]`` (E, in, out), ``b[l]`` (E, 1, out).
Pinned by ``tests/test_ens_parity.py`` against a fixture the unmodified reference's ``fit()`` produced.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch as th
import torch.nn.functional as F

DECAYS = [0.000025, 0.00005, 0.000075, 0.000075, 0.0001]


def forward(W: List[th.Tensor], b: List[th.Tensor], max_lv, min_lv, x, mu=None, sigma=None):
    """``forward(deterministic=True, return_dist=True)`` (:88-121); x (rows, in) or (E, rows, in)."""
    h = (x - mu) / sigma if mu is not None else x
    if h.dim() < 3:
        h = h.unsqueeze(0).repeat(W[0].shape[0], 1, 1)
    for l in range(len(W) - 1):
        h = th.relu(h @ W[l] + b[l])
    out = h @ W[-1] + b[-1]
    mean, logvar = th.chunk(out, 2, dim=-1)
    logvar = max_lv - F.softplus(max_lv - logvar)
    logvar = min_lv + F.softplus(logvar - min_lv)
    return mean, logvar


def loss_fn(W, b, max_lv, min_lv, x, y, mu=None, sigma=None):
    """``_compute_loss`` (:156-169)."""
    mean, logvar = forward(W, b, max_lv, min_lv, x, mu, sigma)
    if y.dim() < 3:
        y = y.unsqueeze(0).repeat(W[0].shape[0], 1, 1)
    total = F.gaussian_nll_loss(mean, y, th.exp(logvar), reduction="none").mean()
    return total + 0.01 * max_lv.sum() - 0.01 * min_lv.sum()


def mse_losses(W, b, max_lv, min_lv, x, y, mu=None, sigma=None):
    """``_compute_mse_losses`` (:171-176)."""
    mean, _ = forward(W, b, max_lv, min_lv, x, mu, sigma)
    y = y.unsqueeze(0).repeat(W[0].shape[0], 1, 1)
    return ((mean - y) ** 2).mean(-1).mean(-1)


def train_step(state: Dict, x, y, step: int, lr: float = 1e-3) -> th.Tensor:
    """One ``optim.step()`` of fit() (:206-212, :248-252).  state: W, b (lists), max_lv, min_lv, mu, sigma, and Adam
    moments 'm' / 'v' as flat lists in the optimiser's parameter order (W0, b0, W1, b1, ..., max_lv, min_lv)."""
    W = [w.detach().clone().requires_grad_(True) for w in state["W"]]
    b = [v.detach().clone().requires_grad_(True) for v in state["b"]]
    mx = state["max_lv"].detach().clone().requires_grad_(True)
    mn = state["min_lv"].detach().clone().requires_grad_(True)
    loss = loss_fn(W, b, mx, mn, x, y, state.get("mu"), state.get("sigma"))
    params = [p for pair in zip(W, b) for p in pair] + [mx, mn]
    grads = th.autograd.grad(loss, params)
    wds = [d for d in DECAYS[:len(W)] for _ in range(2)] + [0.0, 0.0]
    tgt = [p for pair in zip(state["W"], state["b"]) for p in pair] + [state["max_lv"], state["min_lv"]]
    bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
    with th.no_grad():
        for p, g, wd, m, v in zip(tgt, grads, wds, state["m"], state["v"]):
            g = g.add(p, alpha=wd) if wd != 0 else g
            m.lerp_(g, 0.1)
            v.mul_(0.999).addcmul_(g, g, value=0.001)
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(1e-8)
            p.addcdiv_(m, denom, value=-(lr / bc1))
    return loss.detach()


def fit(state: Dict, X: np.ndarray, Y: np.ndarray, *, batch_size, holdout_ratio, max_epochs, normalize, num_elites=2,
        lr=1e-3, max_holdout_size=5000, max_epochs_no_improvement=5):
    """``fit`` (:178-290) with the numpy RNG consumed in the reference's order.  Returns (mean holdout loss, elites)."""
    E = state["W"][0].shape[0]
    if normalize:
        mu, sigma = np.mean(X, axis=0, keepdims=True), np.std(X, axis=0, keepdims=True)
        sigma[sigma < 1e-12] = 1.0
        state["mu"], state["sigma"] = th.tensor(mu).float(), th.tensor(sigma).float()
    tgt = [p for pair in zip(state["W"], state["b"]) for p in pair] + [state["max_lv"], state["min_lv"]]
    state["m"], state["v"] = [th.zeros_like(p) for p in tgt], [th.zeros_like(p) for p in tgt]
    num_holdout = min(int(X.shape[0] * holdout_ratio), max_holdout_size)
    perm = np.random.permutation(X.shape[0])
    inputs, hx = X[perm[num_holdout:]], th.from_numpy(X[perm[:num_holdout]]).float()
    targets, hy = Y[perm[num_holdout:]], th.from_numpy(Y[perm[:num_holdout]]).float()
    idxs = np.random.randint(inputs.shape[0], size=[E, inputs.shape[0]])
    num_batches = int(np.ceil(idxs.shape[-1] / batch_size))
    step, epoch, no_imp = 0, 0, 0
    best = [float("inf")] * E
    hl = None
    while no_imp < max_epochs_no_improvement and epoch < max_epochs:
        for bn in range(num_batches):
            bi = idxs[:, bn * batch_size:(bn + 1) * batch_size]
            step += 1
            train_step(state, th.from_numpy(inputs[bi]).float(), th.from_numpy(targets[bi]).float(), step, lr)
        order = np.argsort(np.random.uniform(size=idxs.shape), axis=-1)
        idxs = idxs[np.arange(idxs.shape[0])[:, None], order]
        hl = [float(v) for v in mse_losses(state["W"], state["b"], state["max_lv"], state["min_lv"], hx, hy,
                                          state.get("mu"), state.get("sigma"))]
        elites = np.argsort(hl)[:num_elites]
        improved = False
        for i in range(E):
            if epoch == 0 or (best[i] - hl[i]) / best[i] > 0.01:
                best[i], no_imp, improved = hl[i], 0, True
        if not improved:
            no_imp += 1
        epoch += 1
    return float(np.mean(hl)), elites
