"""Probabilistic dynamics ensemble (GPI-PD's Dyna model): the oracle and the HIP engine against a fixture produced by
the unmodified reference's ``ProbabilisticEnsemble.fit()`` (9 optimiser steps with per-layer weight decay, holdout
evaluation, elite selection), plus forward / sample outputs.  ``sim`` = wave emulator, ``hip`` = gfx950 (-m gpu)."""
import os

import numpy as np
import pytest
import torch as th

import ens_oracle as eo
from make_golden_ens import CFG, FIT, SEED, data

from morl_baselines_amd.native import load_library

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ens_fit.npz")
NL = len(CFG["arch"]) + 1


@pytest.fixture(scope="module", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    if request.param == "sim":
        import simlib
        return simlib.load_sim(), th.device("cpu")
    return load_library(), th.device("cuda:0")


@pytest.mark.parametrize("norm", [True, False])
def test_oracle_reproduces_reference_fit(norm):
    g = np.load(GOLD)
    tag = f"n{int(norm)}"
    X, Y, probe = data()
    st = dict(W=[th.tensor(g[f"{tag}_init_W{l}"]) for l in range(NL)], b=[th.tensor(g[f"{tag}_init_b{l}"]) for l in range(NL)],
              max_lv=th.ones(1, CFG["output_dim"]) / 2.0, min_lv=-th.ones(1, CFG["output_dim"]) * 10.0)
    np.random.seed(SEED + 1)
    hl, elites = eo.fit(st, X, Y, normalize=norm, **FIT)
    assert hl == pytest.approx(float(g[f"{tag}_holdout"]), rel=1e-5)
    assert list(elites) == list(g[f"{tag}_elites"])
    for l in range(NL):
        np.testing.assert_allclose(st["W"][l].numpy(), g[f"{tag}_W{l}"], rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(st["b"][l].numpy(), g[f"{tag}_b{l}"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(st["max_lv"].numpy(), g[f"{tag}_max_logvar"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("norm", [True, False])
def test_engine_reproduces_reference_fit(be, norm):
    from morl_baselines_amd.dynamics import ProbabilisticEnsemble
    lib, dev = be
    g = np.load(GOLD)
    tag = f"n{int(norm)}"
    X, Y, probe = data()
    th.manual_seed(SEED)
    model = ProbabilisticEnsemble(normalize_inputs=norm, device=dev, lib=lib, max_rows=64, **CFG)
    # construction consumes torch's generator like the reference -> same initial weights; load the fixture's anyway
    sd0 = model.state_dict()
    for l in range(NL):
        np.testing.assert_allclose(sd0[f"layers.{l}.W"].cpu().numpy(), g[f"{tag}_init_W{l}"], rtol=1e-5, atol=1e-6)
    model.load_state_dict({**sd0, **{f"layers.{l}.W": th.tensor(g[f"{tag}_init_W{l}"]) for l in range(NL)},
                           **{f"layers.{l}.b": th.tensor(g[f"{tag}_init_b{l}"]) for l in range(NL)}})
    np.random.seed(SEED + 1)
    hl = model.fit(X, Y, **FIT)
    assert hl == pytest.approx(float(g[f"{tag}_holdout"]), rel=2e-5)
    assert list(model.elites) == list(g[f"{tag}_elites"])
    sd = model.state_dict()
    for l in range(NL):
        np.testing.assert_allclose(sd[f"layers.{l}.W"].cpu().numpy(), g[f"{tag}_W{l}"], rtol=2e-4, atol=5e-6)
        np.testing.assert_allclose(sd[f"layers.{l}.b"].cpu().numpy(), g[f"{tag}_b{l}"], rtol=2e-4, atol=5e-6)
    np.testing.assert_allclose(sd["max_logvar"].cpu().numpy(), g[f"{tag}_max_logvar"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(sd["min_logvar"].cpu().numpy(), g[f"{tag}_min_logvar"], rtol=1e-5, atol=2e-6)
    mean, logvar = model(th.tensor(probe), deterministic=True, return_dist=True)
    np.testing.assert_allclose(mean.cpu().numpy(), g[f"{tag}_probe_mean"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logvar.cpu().numpy(), g[f"{tag}_probe_logvar"], rtol=1e-4, atol=2e-5)
    np.random.seed(SEED + 2)
    s, v, u = model.sample(th.tensor(probe), deterministic=True)
    np.testing.assert_allclose(s, g[f"{tag}_sample"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v, g[f"{tag}_vars"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(u, g[f"{tag}_unc"], rtol=2e-4, atol=1e-6)
    # a loss value of one more step agrees with the oracle's _compute_loss on the same batch
    st = dict(W=[sd[f"layers.{l}.W"].cpu() for l in range(NL)], b=[sd[f"layers.{l}.b"].cpu() for l in range(NL)],
              max_lv=sd["max_logvar"].cpu(), min_lv=sd["min_logvar"].cpu())
    mu = sd.get("inputs_mu"); sg = sd.get("inputs_sigma")
    xb = th.tensor(X[:24]).reshape(3, 8, -1).contiguous()
    yb = th.tensor(Y[:24]).reshape(3, 8, -1).contiguous()
    want = eo.loss_fn(st["W"], st["b"], st["max_lv"], st["min_lv"], xb, yb, None if mu is None else mu.cpu(),
                      None if sg is None else sg.cpu())
    got = model.train_step(xb.to(dev), yb.to(dev), want_loss=True)
    assert float(got) == pytest.approx(float(want), rel=1e-5)
