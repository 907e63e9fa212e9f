"""Shared by the GPI-PD oracle / kernel parity tests."""
from __future__ import annotations

import os

import numpy as np
import torch as th

import gpi_oracle as go
from ac_oracle import clone
from cases_gpi import GpiCase, make_inputs, rows_and_weights, spec_of

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(c: GpiCase):
    return np.load(os.path.join(GOLDEN_DIR, f"gpi_{c.name}.npz"))


def drop_lists(inp):
    T = th.tensor
    return {k: [[T(m) for m in net] for net in v] for k, v in inp["drop"].items()}


def run_oracle(c: GpiCase, inp=None):
    inp = inp or make_inputs(c)
    spec = spec_of(c)
    q, tq = [clone(n) for n in inp["q"]], [clone(n) for n in inp["tq"]]
    state = {k: clone(v) for k, v in inp["state"].items()}
    batch, w, sampled_w = rows_and_weights(c, inp)
    out = go.gpi_update(spec, q, tq, state, batch, w, sampled_w, drop_lists(inp), gamma=c.gamma, lr=c.lr, step=c.step,
                        min_priority=c.min_priority, gpi_pd=c.gpi_pd,
                        max_grad_norm=None if c.max_grad_norm < 0 else c.max_grad_norm, n_per=c.B if c.per else None)
    return dict(q=q, tq=tq, state=state), out


def check_params_against_golden(c: GpiCase, q_after, state, g, grad_tol_frac=None):
    """Parameters / Adam moments after the step vs the reference fixture (see ac_common for the Adam-noise bound)."""
    s = c.subsample
    npar = len(q_after[0])
    bc1, bc2 = 1 - 0.9 ** c.step, 1 - 0.999 ** c.step
    for n in range(2):
        for i in range(npar):
            got = np.asarray(q_after[n][i], dtype=np.float64).reshape(-1)[::s]
            want = g[f"q{n}_{i}"].astype(np.float64)
            m_g, v_g = g[f"q{n}_m_{i}"].astype(np.float64), g[f"q{n}_v_{i}"].astype(np.float64)
            tol = 0.02 * c.lr + 2e-5 * np.abs(want)
            if grad_tol_frac is not None:
                g_scale = np.abs(m_g).max() / (0.1 if c.step == 1 else 1.0) + 1e-30
                tol = tol + np.minimum(2.2 * c.lr, 3.0 * (c.lr / bc1) * grad_tol_frac * g_scale / (np.sqrt(v_g / bc2) + 1e-8))
            err = np.abs(got - want)
            assert (err <= tol).all(), f"q{n}_{i}: max err {err.max():.3e}"
            m = np.asarray(state["exp_avg"][n * npar + i], dtype=np.float64).reshape(-1)[::s]
            m_scale = np.abs(m_g).max() + 1e-30
            assert (np.abs(m - m_g) <= 1e-4 * np.abs(m_g) + 5e-5 * m_scale).all(), f"q{n}_m_{i}: {np.abs(m - m_g).max():.3e}"
