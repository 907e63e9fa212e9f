"""Shared by the actor-critic oracle / kernel parity tests: run the ORACLE on a golden case, compare with a fixture."""
from __future__ import annotations

import os

import numpy as np
import torch as th

import ac_oracle as ac
from cases_ac import ACCase, make_inputs, specs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(c: ACCase):
    return np.load(os.path.join(GOLDEN_DIR, f"ac_{c.name}.npz"))


def gpipd_rows(c: ACCase, inp):
    """The (possibly doubled) batch and per-row weights of gpi_pd_continuous_action.py:381-393."""
    keys = ("obs", "actions", "rewards", "next_obs", "dones")
    batch = [th.tensor(inp[k]) for k in keys]
    if c.n_support > 1:
        batch = [b.repeat(2, 1) for b in batch]
        w = th.vstack([th.tensor(inp["weight"])] * c.B + [th.tensor(inp["support"][i]) for i in inp["choice"][:c.B]])
    else:
        w = th.tensor(inp["weight"]).repeat(c.B, 1)
    return batch, w


def run_oracle(c: ACCase, inp=None):
    """Returns (state dict of parameter lists AFTER the update, oracle output dict)."""
    inp = inp or make_inputs(c)
    qspec, trunk = specs(c)
    st = dict(q=[ac.clone(n) for n in inp["q"]], tq=[ac.clone(n) for n in inp["tq"]], pol=ac.clone(inp["pol"]),
              q_state={k: ac.clone(v) for k, v in inp["q_state"].items()},
              p_state={k: ac.clone(v) for k, v in inp["p_state"].items()})
    T = th.tensor
    if c.algo == "capql":
        batch = (T(inp["obs"]), T(inp["actions"]), T(inp["w"]), T(inp["rewards"]), T(inp["next_obs"]),
                 T(inp["dones"]).reshape(-1))
        out = ac.capql_update(qspec, trunk, st["q"], st["tq"], st["pol"], st["q_state"], st["p_state"], batch,
                              T(inp["eps_next"]), T(inp["eps_pi"][0]), inp["scale"], inp["bias"], gamma=c.gamma,
                              alpha=c.alpha, lr=c.lr, tau=c.tau, step=c.step)
    elif c.algo == "mosac":
        st["log_alpha"] = th.tensor([c.log_alpha0])
        st["al_state"] = {k: ac.clone(v) for k, v in inp["al_state"].items()}
        batch = tuple(T(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones"))
        alpha = float(np.exp(np.float32(c.log_alpha0))) if c.autotune else c.alpha
        if c.autotune:
            alpha = th.tensor([c.log_alpha0]).exp().item()
        out = ac.mosac_update(qspec, trunk, st["q"], st["tq"], st["pol"], st["log_alpha"], st["q_state"],
                              st["p_state"], st["al_state"], batch, T(inp["weights"]), T(inp["eps_next"]),
                              [T(e) for e in inp["eps_pi"]], [T(e) for e in inp["eps_alpha"]], inp["scale"],
                              inp["bias"], gamma=c.gamma, tau=c.tau, q_lr=c.q_lr, policy_lr=c.lr, q_step=c.step,
                              a_step=c.step, policy_freq=c.policy_freq,
                              do_policy=(c.global_step % c.policy_freq == 0), do_target=True, autotune=c.autotune,
                              alpha=alpha, target_entropy=-float(c.Ad))
    elif c.algo == "sacd":
        st["log_alpha"] = th.tensor([c.log_alpha0])
        st["al_state"] = {k: ac.clone(v) for k, v in inp["al_state"].items()}
        batch = tuple(T(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones"))
        alpha = th.tensor([c.log_alpha0]).exp().item() if c.autotune else c.alpha
        te = float(-0.89 * th.log(1 / th.tensor(c.Ad))) if c.autotune else 0.0
        pspec = ac.MlpSpec(c.D, c.arch, c.Ad)
        out = ac.mosac_discrete_update(qspec, pspec, st["q"], st["tq"], st["pol"], st["log_alpha"], st["q_state"],
                                       st["p_state"], st["al_state"], batch, T(inp["weights"]), n_actions=c.Ad,
                                       reward_dim=c.R, gamma=c.gamma, tau=c.tau, q_lr=c.q_lr, policy_lr=c.lr,
                                       step=c.step, do_target=True, autotune=c.autotune, alpha=alpha, target_entropy=te)
    else:
        st["tpol"] = ac.clone(inp["tpol"])
        batch, w = gpipd_rows(c, inp)
        drop = {k: [[T(m) for m in net] for net in v] for k, v in inp["drop"].items()}
        out = ac.gpipd_cont_update(qspec, trunk, st["q"], st["tq"], st["pol"], st["tpol"], st["q_state"],
                                   st["p_state"], batch, w, T(inp["eps_next"]), drop, inp["scale"], inp["bias"],
                                   gamma=c.gamma, lr=c.lr, tau=c.tau, q_step=c.step, p_step=c.step,
                                   do_policy=(c.n_updates % 2 == 0), n_per=(c.B if c.per else None))
    return st, out


def _cmp(name, got, want, sub, rtol, atol):
    got = np.asarray(got, dtype=np.float64).reshape(-1)[::sub]
    want = np.asarray(want, dtype=np.float64).reshape(-1)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} (tol {tol[np.argmax(err - tol)]:.3e})"


def _adam_tol(c, g, key, sub, lr, grad_tol_frac):
    """Per-entry absolute tolerance of a parameter after one Adam step when its gradient is only known to
    ``grad_tol_frac * max|g|``: the step is lr/bc1 * m / (sqrt(v/bc2) + eps), so an entry whose gradient is eps-sized
    (|g| ~ 1e-8) turns a 1e-8 absolute gradient difference into a sizeable fraction of lr.  Bounded by 2.2 lr."""
    m, v = np.abs(g[f"{key.replace('_', '_m_', 1)}"]).astype(np.float64), g[f"{key.replace('_', '_v_', 1)}"].astype(np.float64)
    bc1, bc2 = 1 - 0.9 ** c.step, 1 - 0.999 ** c.step
    g_scale = m.max() / (0.1 if c.step == 1 else 1.0) + 1e-30
    denom = np.sqrt(v / bc2) + 1e-8
    return np.minimum(2.2 * lr, 3.0 * (lr / bc1) * (grad_tol_frac * g_scale) / denom)


def check_against_golden(c: ACCase, st, g, *, q_opt=None, p_opt=None, rtol=2e-5, lr_frac=0.02, grad_tol_frac=None):
    """st: parameter lists after the update; q_opt / p_opt: dicts with exp_avg / exp_avg_sq lists (chained order).
    grad_tol_frac: expected relative (to max|g|) gradient noise of the implementation under test; None = the
    implementation shares the reference's BLAS (oracle) and a small fraction of lr suffices."""
    s = c.subsample
    nq = len(st["q"][0])
    # a first Adam step moves every entry by ~lr * g / (|g| + eps): entries whose gradient is ~eps-sized amplify
    # 1e-9 absolute gradient differences into a fraction of lr, hence the lr-relative absolute term
    q_lr = c.q_lr if c.algo in ("mosac", "sacd") else c.lr
    pa_q, pa_p = lr_frac * q_lr, lr_frac * c.lr

    def tol(key, lr, base):
        if grad_tol_frac is None or f"{key.replace('_', '_m_', 1)}" not in g:
            return base
        return base + _adam_tol(c, g, key, s, lr, grad_tol_frac)

    for n in range(2):
        for i in range(nq):
            t = tol(f"q{n}_{i}", q_lr, pa_q)
            _cmp(f"q{n}_{i}", st["q"][n][i], g[f"q{n}_{i}"], s, rtol, t)
            _cmp(f"tq{n}_{i}", st["tq"][n][i], g[f"tq{n}_{i}"], s, rtol, t)
            if q_opt is not None:
                j = n * nq + i
                m_scale = float(np.abs(g[f"q{n}_m_{i}"]).max()) + 1e-12
                _cmp(f"q{n}_m_{i}", q_opt["exp_avg"][j], g[f"q{n}_m_{i}"], s, 1e-4, 2e-5 * m_scale)
                v_scale = float(np.abs(g[f"q{n}_v_{i}"]).max()) + 1e-30
                _cmp(f"q{n}_v_{i}", q_opt["exp_avg_sq"][j], g[f"q{n}_v_{i}"], s, 2e-4, 2e-5 * v_scale)
    for i in range(len(st["pol"])):
        t = tol(f"pol_{i}", c.lr, pa_p)
        _cmp(f"pol_{i}", st["pol"][i], g[f"pol_{i}"], s, rtol, t)
        if p_opt is not None and f"pol_m_{i}" in g:
            m_scale = float(np.abs(g[f"pol_m_{i}"]).max()) + 1e-12
            _cmp(f"pol_m_{i}", p_opt["exp_avg"][i], g[f"pol_m_{i}"], s, 1e-4, 5e-5 * m_scale)
    if "tpol" in st:
        for i in range(len(st["tpol"])):
            _cmp(f"tpol_{i}", st["tpol"][i], g[f"tpol_{i}"], s, rtol, tol(f"pol_{i}", c.lr, pa_p))
