"""Pins oracle/gpi_oracle.py (GPI-PD with discrete actions: update, gpi_action, max_action, _reset_priorities) to
fixtures produced by the unmodified reference (tests/golden/make_golden_gpi.py)."""
import numpy as np
import pytest
import torch as th

import gpi_oracle as go
from cases_gpi import GPI_CASES, make_inputs, spec_of
from gpi_common import check_params_against_golden, load_golden, run_oracle


@pytest.mark.parametrize("c", GPI_CASES, ids=lambda c: c.name)
def test_oracle_reproduces_reference(c):
    g = load_golden(c)
    st, out = run_oracle(c)
    assert abs(float(out["critic_loss"]) - float(g["critic_loss"])) <= 1e-5 * float(g["critic_loss"])
    check_params_against_golden(c, st["q"], st["state"], g)
    if c.per:
        key = "gpriority" if c.gpi_pd else "priority"
        np.testing.assert_allclose(out[key], g["priority"], rtol=2e-5)
    spec = spec_of(c)
    sup = [th.tensor(s) for s in make_inputs(c)["support"]]
    for k in range(len(g["gpi_actions"])):
        o, w = th.tensor(g["act_obs"][k]), th.tensor(g["act_w"][k])
        a, pi = go.gpi_action(spec, st["q"][0], o, w, sup)
        assert (a, pi) == (int(g["gpi_actions"][k]), int(g["gpi_policies"][k]))
        assert go.max_action(spec, st["q"], o, w) == int(g["max_actions"][k])
    err = go.reset_priority_errors(spec, st["q"], st["tq"], th.tensor(g["rp_obs"]),
                                   th.tensor(g["rp_actions"].astype(np.float32)), th.tensor(g["rp_rewards"]),
                                   th.tensor(g["rp_next_obs"]), th.tensor(g["rp_dones"]), th.tensor(g["rp_w"]), sup,
                                   gamma=c.gamma, gpi_pd=c.gpi_pd)
    pr = err.clamp(min=c.min_priority).pow(0.6).numpy()
    np.testing.assert_allclose(pr, g["rp_priorities"], rtol=2e-5)
