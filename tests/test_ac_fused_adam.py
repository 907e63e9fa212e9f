"""The optimiser step inside the weight-gradient launch of the actor-critic updates (``gemm_wave4_grouped_tn_batched_adam_kernel``,
``csrc/gemm_wave.h: AdamTile``) against the separate ``ac_adam_kernel`` / Polyak launches (``MORL_AC_ADAM_IN_DW=0``): same
function per element, so parameters, Adam moments, target networks and losses must agree to the last bit over consecutive updates
(the second update reads everything the first one's fused step wrote, the actor phase reads the shadow copy it scattered).
The switch is read once per process, hence the sub-processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SNIPPET = r"""
import dataclasses, hashlib, os, sys
import numpy as np, torch as th
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")]
from cases_ac import AC_CASES, make_inputs
import test_ac_kernels_parity as T
if sys.argv[2] == "gpu":
    from morl_baselines_amd.native import load_library
    lib, dev = load_library(), th.device("cuda:0")
else:
    import simlib
    lib, dev = simlib.load_sim(), th.device("cpu")
by_name = {c.name: c for c in AC_CASES}
# more than 32 batch rows: the weight-gradient launch then runs on the split-K wave tiles, the form that can take the step
cases = [dataclasses.replace(by_name["capql_small"], B=40), dataclasses.replace(by_name["mosac_small"], B=48),
         dataclasses.replace(by_name["mosac_noauto_odd"], B=36, global_step=100),
         dataclasses.replace(by_name["gpipd_nopolicy_plain"], B=40, n_updates=2),
         # LayerNorm + Dropout critics (the reference's GPI-PD default): the gains / shifts are stepped by the launch's extra workgroup
         dataclasses.replace(by_name["gpipd_small"], B=40), by_name["gpipd_support_per"]]
if sys.argv[2] == "gpu":
    cases += [by_name["capql_cheetah"], by_name["mosac_hopper"], by_name["gpipd_hopper"], dataclasses.replace(by_name["gpipd_nopolicy_plain"], B=128, arch=(256, 256), n_updates=2)]
# the head's backward pass rides in the chain up to 16 head outputs (8 action dimensions): the boundary and the first case beyond it
cases += [dataclasses.replace(by_name["capql_small"], Ad=8, B=40, seed=41), dataclasses.replace(by_name["capql_small"], Ad=9, B=40, seed=42)]
h = hashlib.sha256()
# a small population (three learners in one call, more than 32 rows each): learner-indexed step counters, bias corrections and
# counter advance of the fused optimiser step
c = dataclasses.replace(by_name["capql_small"], B=40)
inps = [make_inputs(dataclasses.replace(c, seed=c.seed + 100 * k)) for k in range(3)]
import functools
_plain = T.ACEngine
T.ACEngine = functools.partial(_plain, device_steps=True)       # device-resident Adam step counters, as the MORL/D population keeps them
pop = T.build_engine(c, inps, lib, dev)
T.ACEngine = _plain
assert pop.q_steps is not None
stack = lambda key: np.stack([np.asarray(i[key]) for i in inps])
for it in range(2):
    cfg = pop.make_cfg(gamma=c.gamma, tau=c.tau, alpha=c.alpha, q_lr=c.lr, policy_lr=c.lr, q_step=c.step, policy_step=c.step)
    res = pop.update(cfg, obs=stack("obs"), actions=stack("actions"), rewards=stack("rewards"), next_obs=stack("next_obs"),
                     dones=stack("dones"), w=stack("w"), eps_next=stack("eps_next"), eps_pi=np.stack([i["eps_pi"][0] for i in inps]))
    for k in sorted(res):
        h.update(res[k].cpu().numpy().tobytes())
    for name in ("q", "q_target", "q_exp_avg", "q_exp_avg_sq", "pol", "pol_exp_avg", "pol_exp_avg_sq", "q_steps", "pol_steps"):
        h.update(getattr(pop, name).cpu().numpy().tobytes())
    assert pop.q_steps.cpu().tolist() == [it + 1] * 3 and pop.pol_steps.cpu().tolist() == [it + 1] * 3
# ... and a MOSAC population with a learnt entropy coefficient: there the alpha step is the last reader of the counters
c = dataclasses.replace(by_name["mosac_small"], B=48)
inps = [make_inputs(dataclasses.replace(c, seed=c.seed + 100 * k)) for k in range(2)]
T.ACEngine = functools.partial(_plain, device_steps=True)
pop = T.build_engine(c, inps, lib, dev)
T.ACEngine = _plain
for it in range(2):
    cfg = pop.make_cfg(gamma=c.gamma, tau=c.tau, alpha=c.alpha, q_lr=c.q_lr, policy_lr=c.lr, alpha_lr=c.q_lr, q_step=c.step,
                       policy_step=c.step, do_policy=True, policy_iters=c.policy_freq, autotune=c.autotune, target_entropy=-float(c.Ad))
    res = pop.update(cfg, obs=stack("obs"), actions=stack("actions"), rewards=stack("rewards"), next_obs=stack("next_obs"),
                     dones=stack("dones"), w=stack("weights"), eps_next=stack("eps_next"),
                     eps_pi=np.stack([np.stack([i["eps_pi"][k] for i in inps]) for k in range(c.policy_freq)]),
                     eps_alpha=np.stack([np.stack([i["eps_alpha"][k] for i in inps]) for k in range(c.policy_freq)]))
    for k in sorted(res):
        h.update(res[k].cpu().numpy().tobytes())
    for name in ("q", "q_target", "q_exp_avg_sq", "pol", "pol_exp_avg", "log_alpha", "q_steps", "pol_steps"):
        h.update(getattr(pop, name).cpu().numpy().tobytes())
    assert pop.q_steps.cpu().tolist() == [it + 1] * 2 and pop.pol_steps.cpu().tolist() == [(it + 1) * c.policy_freq] * 2, (pop.q_steps, pop.pol_steps)
for c in cases:
    inp = make_inputs(c)
    eng = T.build_engine(c, inp, lib, dev)
    for it in range(2):
        res = T.run_engine(c, inp, eng, ["critic_loss", "policy_loss"])
        for k in sorted(res):
            h.update(res[k].cpu().numpy().tobytes())
        for name in ("q", "q_target", "q_exp_avg", "q_exp_avg_sq", "pol", "pol_exp_avg", "pol_exp_avg_sq", "pol_target", "log_alpha"):
            t = getattr(eng, name, None)
            if t is not None:
                h.update(t.cpu().numpy().tobytes())
print("AC_DIGEST", h.hexdigest())
"""


def _digest(mode, extra_env):
    r = subprocess.run([sys.executable, "-c", _SNIPPET, ROOT, mode], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, **extra_env), cwd=ROOT)
    assert r.returncode == 0 and "AC_DIGEST" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout.split("AC_DIGEST")[1].strip()


def test_adam_inside_the_weight_gradient_launch_gives_the_bits_of_the_separate_launches():
    assert _digest("sim", {}) == _digest("sim", {"MORL_AC_ADAM_IN_DW": "0"})


def test_paired_policy_heads_give_the_bits_of_the_separate_launches():
    """a ~ pi(s) sampled next to a' ~ pi(s') (``ac_head_fwd_pair_kernel``, critic input rows of its own for the actor phase) against
    one head launch per phase (``MORL_AC_HEADS_PAIRED=0``)."""
    assert _digest("sim", {}) == _digest("sim", {"MORL_AC_HEADS_PAIRED": "0", "MORL_AC_ADAM_IN_DW": "0"})


def test_head_backward_inside_the_actor_chain_gives_the_bits_of_the_separate_launch():
    """dLoss/d(head pre-activations) computed by the input stage of the actor's backward chain (``mlp_chain16_headbwd_kernel``,
    the same ``ac_head_bwd_row``) against ``ac_head_bwd_kernel`` as its own launch (``MORL_AC_HEADBWD_IN_CHAIN=0``)."""
    assert _digest("sim", {}) == _digest("sim", {"MORL_AC_HEADBWD_IN_CHAIN": "0"})


@pytest.mark.gpu
def test_adam_inside_the_weight_gradient_launch_gives_the_bits_of_the_separate_launches_on_the_gpu():
    assert _digest("gpu", {}) == _digest("gpu", {"MORL_AC_ADAM_IN_DW": "0"})
    assert _digest("gpu", {}) == _digest("gpu", {"MORL_AC_HEADS_PAIRED": "0"})
    assert _digest("gpu", {}) == _digest("gpu", {"MORL_AC_HEADBWD_IN_CHAIN": "0"})
