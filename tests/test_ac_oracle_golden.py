"""Pins oracle/ac_oracle.py (CAPQL / MOSAC / GPI-PD-continuous updates) to fixtures produced by the unmodified
reference (tests/golden/make_golden_ac.py): losses to 1e-5 relative, parameters / Adam moments after the step."""
import numpy as np
import pytest

from ac_common import check_against_golden, load_golden, run_oracle
from cases_ac import AC_CASES


@pytest.mark.parametrize("c", AC_CASES, ids=lambda c: c.name)
def test_oracle_reproduces_reference_update(c):
    g = load_golden(c)
    st, out = run_oracle(c)
    check_against_golden(c, st, g, q_opt=st["q_state"], p_opt=st["p_state"])
    rel = lambda a, b: abs(float(a) - float(b)) <= 1e-5 * max(abs(float(b)), 1e-3)  # noqa: E731
    if c.algo == "capql":
        assert rel(out["critic_loss"], g["critic_loss"]) and rel(out["policy_loss"], g["policy_loss"])
    elif c.algo == "mosac":
        if "qf1_loss" in g:
            assert rel(out["qf1_loss"], g["qf1_loss"]) and rel(out["qf2_loss"], g["qf2_loss"])
            assert rel(out["actor_losses"][-1], g["actor_loss"])
            if c.autotune:
                assert rel(out["alpha_losses"][-1], g["alpha_loss"])
        assert rel(out["alpha"], g["alpha"])
        if "log_alpha" in g:
            np.testing.assert_allclose(st["log_alpha"].numpy(), g["log_alpha"], rtol=1e-5, atol=1e-7)
    elif c.algo == "sacd":
        assert rel(out["qf1_loss"], g["qf1_loss"]) and rel(out["qf2_loss"], g["qf2_loss"])
        assert rel(out["actor_loss"], g["actor_loss"]) and rel(out["alpha"], g["alpha"])
        if c.autotune:
            assert rel(out["alpha_loss"], g["alpha_loss"])
            np.testing.assert_allclose(st["log_alpha"].numpy(), g["log_alpha"], rtol=1e-5, atol=1e-7)
    else:
        if "critic_loss" in g:
            assert rel(out["critic_loss"], g["critic_loss"]) and rel(out["policy_loss"], g["policy_loss"])
        if c.per:
            np.testing.assert_allclose(out["priority"], g["priority"], rtol=1e-5)
