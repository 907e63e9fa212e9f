"""GPI-PD (discrete actions) through the C ABI (morl_gpi_*) against the oracle and the reference-generated fixtures:
update (losses 1e-5, gradients, parameters after the step, PER errors), GPI / greedy actions (exact indices) and the
_reset_priorities errors.  ``sim`` = host wave emulator, ``hip`` = gfx950 (-m gpu)."""
import numpy as np
import pytest
import torch as th

import gpi_oracle as go
from cases_gpi import GPI_CASES, make_inputs, rows_and_weights, spec_of
from gpi_common import check_params_against_golden, load_golden, run_oracle

from morl_baselines_amd.gpi_engine import GPIEngine
from morl_baselines_amd.native import load_library


@pytest.fixture(scope="module", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    if request.param == "sim":
        import simlib
        return simlib.load_sim(), th.device("cpu")
    return load_library(), th.device("cuda:0")


def build(c, inp, lib, dev):
    K = inp["K"]
    eng = GPIEngine(c.D, c.A, c.R, c.arch, max_rows=max(2 * c.B, 20), max_support=max(K, c.n_support, 1), layer_norm=c.layer_norm,
                    drop_rate=c.drop_rate, device=dev, lib=lib)
    npar = len(inp["q"][0])
    with th.no_grad():
        for n in range(2):
            for v, s in zip(eng.views(eng.q, n), inp["q"][n]):
                v.copy_(s)
            for v, s in zip(eng.views(eng.q_target, n), inp["tq"][n]):
                v.copy_(s)
            for v, s in zip(eng.views(eng.exp_avg, n), inp["state"]["exp_avg"][n * npar:(n + 1) * npar]):
                v.copy_(s)
            for v, s in zip(eng.views(eng.exp_avg_sq, n), inp["state"]["exp_avg_sq"][n * npar:(n + 1) * npar]):
                v.copy_(s)
    return eng


def pack_masks(c, inp, dev):
    if not inp["drop"]:
        return None
    parts = []
    for key in ("target",) + (("env",) if c.gpi_pd else ()) + ("q",):
        for n in range(2):
            for m in inp["drop"][key][n]:
                parts.append(np.ascontiguousarray(m, dtype=np.uint8).reshape(-1))
    return th.tensor(np.concatenate(parts)).to(dev)


def close(a, b, rtol, frac):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=frac * (np.abs(b).max() + 1e-30))


def run_and_check_against_oracle(lib, dev, c):
    """One ``GPIPD.update`` body of case ``c`` through the C ABI against the oracle on the same inputs; returns
    (engine, inputs, device results)."""
    inp = make_inputs(c)
    eng = build(c, inp, lib, dev)
    batch, w, sampled_w = rows_and_weights(c, inp)
    want = ["critic_loss", "td_error", "target_q", "grads"] + (["gtd_error", "target_q_envelope"] if c.gpi_pd else []) + \
        (["grad_norm"] if c.max_grad_norm >= 0 else [])
    res = eng.update(obs=batch[0], actions=batch[1], rewards=batch[2], next_obs=batch[3], dones=batch[4], w=w,
                     sampled_w=sampled_w if c.gpi_pd else None, gamma=c.gamma, lr=c.lr, adam_step=c.step,
                     min_priority=c.min_priority, max_grad_norm=None if c.max_grad_norm < 0 else c.max_grad_norm,
                     gpi_pd=c.gpi_pd, n_per=c.B, drop_masks=pack_masks(c, inp, dev), want=want)
    res = {k: v.cpu() for k, v in res.items()}
    st, out = run_oracle(c, inp)
    # ---- oracle ---------------------------------------------------------------------------------------------------------
    assert abs(float(res["critic_loss"]) - float(out["critic_loss"])) <= 1e-5 * float(out["critic_loss"])
    close(res["target_q"], out["target_q"], 1e-5, 1e-6)
    close(res["td_error"], out["td_error"], 1e-5, 1e-6)
    if c.gpi_pd:
        close(res["target_q_envelope"], out["target_env"], 1e-5, 1e-6)
        close(res["gtd_error"], out["gtd_error"], 1e-5, 1e-6)
    npar = len(inp["q"][0])
    shapes = eng.shapes()
    for n in range(2):
        o = 0
        for i, s in enumerate(shapes):
            k = int(np.prod(s))
            close(res["grads"][n, o:o + k].view(s), out["grads"][n * npar + i], 2e-4, 3e-5)
            o += k
    if c.max_grad_norm >= 0:
        close(res["grad_norm"], th.stack(out["norms"]), 1e-5, 1e-6)
    return eng, inp, res


@pytest.mark.parametrize("c", GPI_CASES, ids=lambda c: c.name)
def test_update_actions_priorities(be, c):
    lib, dev = be
    if dev.type == "cpu" and max(c.arch) >= 256:
        pytest.skip("reference-sized networks run on the GPU only (the emulator is slow)")
    eng, inp, res = run_and_check_against_oracle(lib, dev, c)
    # ---- the fixture of the unmodified reference -------------------------------------------------------------------------
    g = load_golden(c)
    assert abs(float(res["critic_loss"]) - float(g["critic_loss"])) <= 1e-5 * float(g["critic_loss"])
    q_after = [[v.cpu() for v in eng.views(eng.q, n)] for n in range(2)]
    state = dict(exp_avg=[v.cpu() for n in range(2) for v in eng.views(eng.exp_avg, n)])
    check_params_against_golden(c, q_after, state, g, grad_tol_frac=3e-6)
    pr = (res["gtd_error"] if c.gpi_pd else res["td_error"]).numpy().clip(min=c.min_priority) ** 0.6
    np.testing.assert_allclose(pr, g["priority"], rtol=3e-5)
    # ---- action selection: exact indices ----------------------------------------------------------------------------------
    sup = th.tensor(inp["support"])
    for k in range(len(g["gpi_actions"])):
        a = eng.action(g["act_obs"][k], g["act_w"][k], sup).cpu().tolist()
        assert a == [int(g["gpi_actions"][k]), int(g["gpi_policies"][k])], k
        assert int(eng.action(g["act_obs"][k], g["act_w"][k], None)[0]) == int(g["max_actions"][k]), k
    # ---- _reset_priorities errors ------------------------------------------------------------------------------------------
    err = eng.priority_errors(g["rp_obs"], g["rp_actions"], g["rp_rewards"], g["rp_next_obs"], g["rp_dones"], g["rp_w"],
                              sup, gamma=c.gamma, gpi_pd=c.gpi_pd).cpu()
    np.testing.assert_allclose(err.clamp(min=c.min_priority).pow(0.6).numpy(), g["rp_priorities"], rtol=5e-5)
    # eval-mode Q of the updated ensemble == oracle forward
    q = eng.q_forward(g["act_obs"], g["act_w"], nets=2).cpu()
    spec = spec_of(c)
    for n in range(2):
        want_q = go.qnet_forward(spec, q_after[n], th.tensor(g["act_obs"]), th.tensor(g["act_w"]))
        close(q[n], want_q, 1e-5, 2e-6)
