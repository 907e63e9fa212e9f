"""LayerNorm / Dropout networks (GPI-PD's critics: ``gpi_pd.py:41-76``, ``gpi_pd_continuous_action.py:60-73``; ``common/networks.py:10-48``)
on the layer-fused 16-row chain whose hidden steps carry the post-op (``csrc/mlp_chain16.h``: ``mlp_chain16_post_kernel``) against
the per-layer launches (a GEMM + a post-op launch per layer forward; a dX GEMM + a LayerNorm-gradient + a post-op launch per layer
backward): ``MORL_AC_LN_CHAIN=3`` (default) vs ``0``, each in its own process (the switch is read once).

* the post-op stages repeat the per-layer kernels' arithmetic in the same order, so everything downstream of a Linear output is
  bit-identical GIVEN that output; the Linear outputs themselves come from differently ordered exact-fp32 sums (k-ordered chain vs
  split-K wave tiles): results agree to fp32 rounding (1e-5 of the largest entry), and each leg passes the reference fixture on
  its own (tests/test_gpi_kernels_parity.py, tests/test_ac_kernels_parity.py run under both settings here);
* the keep masks of the counter-based Dropout are the same bits in both legs (same hash, same indices);
* the launch count drops (counted by the emulator): the test fails if the chain silently stops applying."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SNIPPET = r"""
import ctypes, os, sys, pickle
import numpy as np, torch as th
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")]
if sys.argv[2] == "sim":
    import simlib
    lib, dev = simlib.load_sim(), th.device("cpu")
    count = lib.lib.hipsim_launch_count
    count.restype = ctypes.c_longlong
else:
    from morl_baselines_amd.native import load_library
    lib, dev = load_library(), th.device("cuda:0")
    count = lambda: 0                                   # (launches are counted by the emulator only)
out = {}
# (1) GPI-PD, discrete actions: LayerNorm + Dropout trunk [256] x 4, explicit masks of the fixture (reference parity inside)
from cases_gpi import GPI_CASES, GpiCase
import test_gpi_kernels_parity as G
# (the emulator gets a narrower trunk than the reference fixture's [256] x 4 -- same kernels, a fifth of the arithmetic; the oracle
# is the yardstick either way)
c = GpiCase("ln_chain_sim", D=7, A=6, R=3, arch=(64, 64, 64, 64), B=16, n_support=3, seed=11) if sys.argv[2] == "sim" else \
    [c for c in GPI_CASES if c.name == "gpi_minecart"][0]
c0 = count()
eng, inp, res = G.run_and_check_against_oracle(lib, dev, c)
out["gpi_launches"] = count() - c0
out["gpi"] = {k: v.cpu().numpy() for k, v in res.items()}
out["gpi_q"] = eng.q.clone().cpu().numpy()
# the same update with the counter-based Dropout (no explicit masks): the hash, seeds and indices of both paths must agree
from cases_gpi import rows_and_weights
batch, w, sampled_w = rows_and_weights(c, inp)
eng2 = G.build(c, inp, lib, dev)
r2 = eng2.update(obs=batch[0], actions=batch[1], rewards=batch[2], next_obs=batch[3], dones=batch[4], w=w, sampled_w=sampled_w,
                 gamma=c.gamma, lr=c.lr, adam_step=c.step, min_priority=c.min_priority, max_grad_norm=None, gpi_pd=True, n_per=c.B,
                 dropout_seed=1234, want=["critic_loss", "td_error", "grads"])
out["gpi_rng"] = {k: v.cpu().numpy() for k, v in r2.items()}
# (2) GPI-PD, continuous actions (TD3-style learner, LayerNorm + Dropout critics)
from cases_ac import AC_CASES, make_inputs
import test_ac_kernels_parity as T
for c in AC_CASES:
    if c.name.startswith("gpipd") and not (sys.argv[2] == "sim" and c.name == "gpipd_hopper"):
        inp = make_inputs(c)
        eng = T.build_engine(c, inp, lib, dev)
        c0 = count()
        r = T.run_engine(c, inp, eng, ["critic_loss", "policy_loss", "q_grads", "pol_grads"])
        out[c.name + "_launches"] = count() - c0
        out[c.name] = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in r.items()}
sys.stdout.buffer.write(b"PICKLE" + pickle.dumps(out))
"""


def _run(mode, backend):
    r = subprocess.run([sys.executable, "-c", _SNIPPET, ROOT, backend], capture_output=True, timeout=1500,
                       env=dict(os.environ, MORL_AC_LN_CHAIN=str(mode)), cwd=ROOT)
    assert r.returncode == 0 and b"PICKLE" in r.stdout, r.stdout[-2000:].decode(errors="replace") + r.stderr[-3000:].decode(errors="replace")
    import pickle
    return pickle.loads(r.stdout.split(b"PICKLE", 1)[1])


@pytest.fixture(scope="module", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def legs(request):
    return _run(3, request.param), _run(0, request.param), request.param


def _close(a, b, tol=1e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30), np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_chain_and_per_layer_launches_agree(legs):
    fused, sep, _ = legs
    for key in ("gpi", "gpi_rng"):
        for k in fused[key]:
            _close(fused[key][k], sep[key][k])
    _close(fused["gpi_q"], sep["gpi_q"], tol=2e-5)
    for name in fused:
        if name.startswith("gpipd") and not name.endswith("_launches"):
            for k in fused[name]:
                _close(fused[name][k], sep[name][k], tol=3e-5)


def test_the_chain_applies(legs):
    fused, sep, backend = legs
    if backend != "sim":
        pytest.skip("launches are counted by the emulator")
    # GPI-PD discrete: three forward groups (target nets on the rows; on rows x support; online nets) of 3 x (GEMM + post-op) + 1 and a
    # backward of 3 x (dX GEMM + LayerNorm gradient + post-op) + 1 become one launch per pass (+ one LayerNorm-gradient launch)
    print("launches per GPI-PD update:", fused["gpi_launches"], "fused,", sep["gpi_launches"], "per layer")
    assert fused["gpi_launches"] <= sep["gpi_launches"] - 15
    # the continuous learner: the chain takes the LayerNorm / Dropout critics whose hidden layers are wider than 32 (gpipd_small,
    # gpipd_hopper); narrow nets (gpipd_support_per: [48, 32]) and plain ones keep what they ran before
    for name in fused:
        if name.endswith("_launches") and name.startswith("gpipd"):
            print(name, fused[name], sep[name])
            if name[:-9] in ("gpipd_small", "gpipd_hopper"):
                assert fused[name] <= sep[name] - 8
            else:
                assert fused[name] == sep[name]
