"""Golden fixtures of GPI-PD (discrete actions) from the UNMODIFIED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_gpi.py

One ``GPIPD.update(weight)`` per case with the batch (instance attribute), ``random.choices`` / ``random.sample``
(module attributes, restored), the dropout masks (``torch.nn.functional.dropout``) and the logged losses pinned from
outside; plus ``gpi_action`` / ``max_action`` / ``_reset_priorities`` known answers.  Outputs ``tests/golden/gpi_<case>.npz``.
"""
from __future__ import annotations

import os
import random
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
from cases_gpi import GPI_CASES, GpiCase, make_inputs  # noqa: E402
from make_golden_ac import Pins, _dump, _load, _seed_opt  # noqa: E402


def run_case(mod, c: GpiCase) -> dict:
    inp = make_inputs(c)
    env = rh.FakeEnv(c.D, c.A, c.R, env_id="fake-minecart-v0")
    ag = mod.GPIPD(env, learning_rate=c.lr, net_arch=list(c.arch), batch_size=c.B, gamma=c.gamma,
                   max_grad_norm=None if c.max_grad_norm < 0 else c.max_grad_norm, per=c.per, gpi_pd=c.gpi_pd, dyna=False,
                   min_priority=c.min_priority, drop_rate=c.drop_rate, layer_norm=c.layer_norm, log=False, seed=0,
                   device="cpu", buffer_size=64, target_net_update_freq=10 ** 9)
    for n in range(2):
        _load(ag.q_nets[n], inp["q"][n])
        _load(ag.target_q_nets[n], inp["tq"][n])
    qp = [p for net in ag.q_nets for p in net.parameters()]
    _seed_opt(ag.q_optim, qp, inp["state"], c.step)
    T = th.tensor
    batch = (T(inp["obs"]), T(inp["actions"]), T(inp["rewards"]), T(inp["next_obs"]), T(inp["dones"]))
    if c.per:
        batch = batch + (np.arange(c.B),)
    ag._sample_batch_experiences = lambda: batch
    rec = {}
    ag.replay_buffer.update_priorities = lambda idx, pr: rec.__setitem__("priority", np.asarray(pr, np.float64).copy())
    ag.weight_support = [T(s) for s in inp["support"]]
    ag.global_step, ag.log = 100, True
    drops = []
    if c.drop_rate > 0:
        for key in ("target",) + (("env",) if c.gpi_pd else ()) + ("q",):
            for n in range(2):
                drops += inp["drop"][key][n]
    saved_sample = random.sample
    random.sample = lambda pop, k: [pop[i] for i in inp["sample4"][:k]]
    try:
        with Pins(dropout=drops, choices=list(inp["choice"])) as pins:
            ag.update(T(inp["weight"]))
    finally:
        random.sample = saved_sample
    out = dict(critic_loss=np.float32(pins.logged["losses/critic_loss"]))
    if "priority" in rec:
        out["priority"] = rec["priority"]
    for n in range(2):
        _dump(out, f"q{n}", list(ag.q_nets[n].parameters()), ag.q_optim, c.subsample)
    # action selection / priority reset known answers on the UPDATED nets (eval mode -> no dropout draws)
    rng = np.random.default_rng(77 + c.seed)
    obs_a = rng.standard_normal((6, c.D)).astype(np.float32)
    w_a = np.abs(rng.standard_normal((6, c.R))).astype(np.float32)
    w_a /= w_a.sum(1, keepdims=True)
    for net in ag.q_nets + ag.target_q_nets:
        net.eval()
    acts, pis, maxa = [], [], []
    for o, w in zip(obs_a, w_a):
        a, pi = ag.gpi_action(T(o), T(w), return_policy_index=True)
        acts.append(a); pis.append(pi); maxa.append(ag.max_action(T(o), T(w)))
    out.update(act_obs=obs_a, act_w=w_a, gpi_actions=np.asarray(acts), gpi_policies=np.asarray(pis),
               max_actions=np.asarray(maxa))
    # _reset_priorities on a small buffer
    nb = 20
    for _ in range(nb):
        ag.replay_buffer.add(rng.standard_normal(c.D).astype(np.float32), rng.integers(c.A),
                             rng.standard_normal(c.R).astype(np.float32), rng.standard_normal(c.D).astype(np.float32),
                             rng.random() < 0.2)
    del ag.replay_buffer.update_priorities
    wr = T(w_a[0])
    ag._reset_priorities(wr)
    tree = ag.replay_buffer.tree
    out.update(rp_obs=ag.replay_buffer.obs[:nb].copy(), rp_actions=ag.replay_buffer.actions[:nb].copy(),
               rp_rewards=ag.replay_buffer.rewards[:nb].copy(), rp_next_obs=ag.replay_buffer.next_obs[:nb].copy(),
               rp_dones=ag.replay_buffer.dones[:nb].copy(), rp_w=w_a[0], rp_priorities=tree.nodes[-1][:nb].copy())
    return out


def main():
    rh.install_stubs()
    from morl_baselines.multi_policy.gpi_pd import gpi_pd as mod
    th.set_num_threads(1)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for c in GPI_CASES:
        if only and c.name not in only:
            continue
        out = run_case(mod, c)
        np.savez_compressed(os.path.join(HERE, f"gpi_{c.name}.npz"), **out)
        print(c.name, "critic_loss", float(out["critic_loss"]), "actions", out["gpi_actions"].tolist(), out["max_actions"].tolist())


if __name__ == "__main__":
    main()
