"""Golden fixtures of the actor-critic updates, produced by the UNMODIFIED reference in the build container.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_ac.py

For each case of ``cases_ac.py`` the reference's real agent (``CAPQL``, ``MOSAC``, ``GPIPDContinuousAction``) is
constructed on a spaces-only environment, the seeded parameters / optimiser states are loaded, and ONE ``update()`` is
executed with its inputs pinned from outside: the sampled batch (instance attribute), the random draws
(``torch.distributions.normal._standard_normal``, ``torch.randn_like``, ``torch.nn.functional.dropout``,
``random.choices`` -- module attributes, restored afterwards) and the logged losses (the ``wandb`` stand-in's ``log``).
No reference source is modified.  Outputs: ``tests/golden/ac_<case>.npz``.
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
from cases_ac import AC_CASES, ACCase, make_inputs, specs  # noqa: E402


class Pins:
    """Temporarily replaces torch's random draws by queued arrays and records wandb.log payloads."""

    def __init__(self, normal=(), randn_like=(), dropout=(), choices=None):
        self.normal, self.randn, self.drop = list(normal), list(randn_like), list(dropout)
        self.choices = choices
        self.logged = {}

    def __enter__(self):
        import torch.distributions.normal as tdn
        import torch.nn.functional as F
        import wandb

        self._saved = (tdn._standard_normal, th.randn_like, F.dropout, random.choices, wandb.log)

        def std_normal(shape, dtype=None, device=None):
            a = self.normal.pop(0)
            assert tuple(shape) == tuple(a.shape), (shape, a.shape)
            return th.tensor(a)

        def randn_like(x, **k):
            a = self.randn.pop(0)
            assert tuple(x.shape) == tuple(a.shape)
            return th.tensor(a)

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            a = self.drop.pop(0)
            assert x.numel() == a.size and x.shape[-1] == a.shape[-1], (x.shape, a.shape)
            return x * th.tensor(a).reshape(x.shape) * (1.0 / (1.0 - p))

        tdn._standard_normal, th.randn_like, F.dropout = std_normal, randn_like, dropout
        if self.choices is not None:
            random.choices = lambda pop, k=1, **kw: [pop[i] for i in self.choices[:k]]
        wandb.log = lambda d=None, **k: self.logged.update(d or {})
        return self

    def __exit__(self, *a):
        import torch.distributions.normal as tdn
        import torch.nn.functional as F
        import wandb

        tdn._standard_normal, th.randn_like, F.dropout, random.choices, wandb.log = self._saved
        assert not self.normal and not self.randn and not self.drop, "unused pinned draws"


def _load(module, params):
    with th.no_grad():
        ps = list(module.parameters())
        assert len(ps) == len(params), (len(ps), len(params))
        for p, v in zip(ps, params):
            assert p.shape == v.shape, (p.shape, v.shape)
            p.copy_(v)


def _seed_opt(optim, params, state, step):
    if step <= 1:
        return
    for p, m, v in zip(params, state["exp_avg"], state["exp_avg_sq"]):
        optim.state[p] = {"step": th.tensor(float(step - 1)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}


def _dump(out, prefix, params, optim=None, sub=1):
    for i, p in enumerate(params):
        out[f"{prefix}_{i}"] = p.detach().numpy().reshape(-1)[::sub].copy()
        if optim is not None:
            st = optim.state[p]
            out[f"{prefix}_m_{i}"] = st["exp_avg"].numpy().reshape(-1)[::sub].copy()
            out[f"{prefix}_v_{i}"] = st["exp_avg_sq"].numpy().reshape(-1)[::sub].copy()


def _env(c: ACCase):
    env = rh.FakeEnv(c.D, 1, c.R, env_id="fake-halfcheetah-v0", act_dim=c.Ad)
    env.action_space = rh._Box(c.low, c.high, (c.Ad,))
    return env


def _t(a):
    return th.tensor(a)


def run_capql(ref, c: ACCase) -> dict:
    inp = make_inputs(c)
    ag = ref.capql.CAPQL(_env(c), learning_rate=c.lr, gamma=c.gamma, tau=c.tau, net_arch=list(c.arch),
                         batch_size=c.B, alpha=c.alpha, log=False, seed=0, device="cpu")
    for n in range(2):
        _load(ag.q_nets[n], inp["q"][n])
        _load(ag.target_q_nets[n], inp["tq"][n])
    _load(ag.policy, inp["pol"])
    qp = [p for net in ag.q_nets for p in net.parameters()]
    _seed_opt(ag.q_optim, qp, inp["q_state"], c.step)
    _seed_opt(ag.policy_optim, list(ag.policy.parameters()), inp["p_state"], c.step)
    batch = tuple(_t(inp[k]) for k in ("obs", "actions", "w", "rewards", "next_obs", "dones"))
    batch = batch[:5] + (batch[5].reshape(-1),)          # ReplayMemory stacks scalar dones -> (B,)
    ag.replay_buffer.sample = lambda *a, **k: batch
    ag.log, ag.global_step = True, 100
    with Pins(normal=[inp["eps_next"], inp["eps_pi"][0]]) as pins:
        ag.update()
    out = dict(critic_loss=np.float32(pins.logged["losses/critic_loss"]),
               policy_loss=np.float32(pins.logged["losses/policy_loss"]))
    for n in range(2):
        _dump(out, f"q{n}", list(ag.q_nets[n].parameters()), ag.q_optim, c.subsample)
        _dump(out, f"tq{n}", list(ag.target_q_nets[n].parameters()), None, c.subsample)
    _dump(out, "pol", list(ag.policy.parameters()), ag.policy_optim, c.subsample)
    return out


def run_mosac(ref, c: ACCase) -> dict:
    inp = make_inputs(c)
    ag = ref.mosac.MOSAC(_env(c), weights=inp["weights"].copy(), gamma=c.gamma, tau=c.tau, batch_size=c.B,
                         net_arch=list(c.arch), policy_lr=c.lr, q_lr=c.q_lr, policy_freq=c.policy_freq,
                         alpha=c.alpha, autotune=c.autotune, log=False, seed=0, device="cpu", buffer_size=64)
    _load(ag.qf1, inp["q"][0]); _load(ag.qf2, inp["q"][1])
    _load(ag.qf1_target, inp["tq"][0]); _load(ag.qf2_target, inp["tq"][1])
    _load(ag.actor, inp["pol"])
    qp = list(ag.qf1.parameters()) + list(ag.qf2.parameters())
    _seed_opt(ag.q_optimizer, qp, inp["q_state"], c.step)
    _seed_opt(ag.actor_optimizer, list(ag.actor.parameters()), inp["p_state"], c.step)
    if c.autotune:
        with th.no_grad():
            ag.log_alpha.fill_(c.log_alpha0)
        ag.alpha = ag.log_alpha.exp().item()
        ag.alpha_tensor = th.scalar_tensor(ag.alpha)
        _seed_opt(ag.a_optimizer, [ag.log_alpha], inp["al_state"], c.step)
    batch = tuple(_t(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones")) + (None,)
    ag.buffer.sample = lambda *a, **k: batch
    ag.global_step = c.global_step
    ag.log = (c.global_step % 100 == 0)
    do_policy = c.global_step % c.policy_freq == 0
    draws = [inp["eps_next"]]
    if do_policy:
        for k in range(c.policy_freq):
            draws.append(inp["eps_pi"][k])
            if c.autotune:
                draws.append(inp["eps_alpha"][k])
    with Pins(normal=draws) as pins:
        ag.update()
    out = dict(alpha=np.float64(ag.alpha))
    for k, v in pins.logged.items():
        if k.startswith("losses/"):
            out[k.split("/")[1]] = np.float64(v)
    _dump(out, "q0", list(ag.qf1.parameters()), ag.q_optimizer, c.subsample)
    _dump(out, "q1", list(ag.qf2.parameters()), ag.q_optimizer, c.subsample)
    _dump(out, "tq0", list(ag.qf1_target.parameters()), None, c.subsample)
    _dump(out, "tq1", list(ag.qf2_target.parameters()), None, c.subsample)
    if do_policy:
        _dump(out, "pol", list(ag.actor.parameters()), ag.actor_optimizer, c.subsample)
    else:
        _dump(out, "pol", list(ag.actor.parameters()), None, c.subsample)
    if c.autotune and do_policy:
        out["log_alpha"] = ag.log_alpha.detach().numpy().copy()
    return out


def run_sacd(ref, c: ACCase) -> dict:
    inp = make_inputs(c)
    env = rh.FakeEnv(c.D, c.Ad, c.R, env_id="fake-minecart-v0")
    ag = ref.sacd.MOSACDiscrete(env, weights=inp["weights"].copy(), gamma=c.gamma, tau=c.tau, batch_size=c.B,
                                net_arch=list(c.arch), policy_lr=c.lr, q_lr=c.q_lr, alpha=c.alpha, autotune=c.autotune,
                                target_net_freq=4, update_frequency=4, log=False, seed=0, device="cpu", buffer_size=64)
    _load(ag.qf1, inp["q"][0]); _load(ag.qf2, inp["q"][1])
    _load(ag.qf1_target, inp["tq"][0]); _load(ag.qf2_target, inp["tq"][1])
    _load(ag.actor, inp["pol"])
    _seed_opt(ag.q_optimizer, list(ag.qf1.parameters()) + list(ag.qf2.parameters()), inp["q_state"], c.step)
    _seed_opt(ag.actor_optimizer, list(ag.actor.parameters()), inp["p_state"], c.step)
    if c.autotune:
        with th.no_grad():
            ag.log_alpha.fill_(c.log_alpha0)
        ag.alpha = ag.log_alpha.exp().item()
        ag.alpha_tensor = th.scalar_tensor(ag.alpha)
        _seed_opt(ag.a_optimizer, [ag.log_alpha], inp["al_state"], c.step)
    batch = tuple(_t(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones")) + (None,)
    ag.buffer.sample = lambda *a, **k: batch
    ag.global_step, ag.log = 100, True          # % target_net_freq == 0 -> polyak; % 100 == 0 -> losses are logged
    with Pins() as pins:
        ag.update()
    out = dict(alpha=np.float64(ag.alpha), target_entropy=np.float64(float(ag.target_entropy)) if c.autotune else np.float64(0))
    for k, v in pins.logged.items():
        if k.startswith("losses/"):
            out[k.split("/")[1]] = np.float64(v)
    _dump(out, "q0", list(ag.qf1.parameters()), ag.q_optimizer, c.subsample)
    _dump(out, "q1", list(ag.qf2.parameters()), ag.q_optimizer, c.subsample)
    _dump(out, "tq0", list(ag.qf1_target.parameters()), None, c.subsample)
    _dump(out, "tq1", list(ag.qf2_target.parameters()), None, c.subsample)
    _dump(out, "pol", list(ag.actor.parameters()), ag.actor_optimizer, c.subsample)
    if c.autotune:
        out["log_alpha"] = ag.log_alpha.detach().numpy().copy()
    return out


def run_gpipd(ref, c: ACCase) -> dict:
    inp = make_inputs(c)
    qspec, _ = specs(c)
    mod = ref.gpipd_cont
    ag = mod.GPIPDContinuousAction(_env(c), learning_rate=c.lr, gamma=c.gamma, tau=c.tau, net_arch=list(c.arch),
                                   batch_size=c.B, gradient_updates=1, per=c.per, dyna=False, log=False, seed=0,
                                   device="cpu", buffer_size=64)
    if not (c.layer_norm and c.drop_rate == 0.01):       # non-default Q-net flavour: rebuild with the module's class
        for lst in (ag.q_nets, ag.target_q_nets):
            for n in range(2):
                lst[n] = mod.QNetwork(c.D, c.Ad, c.R, net_arch=list(c.arch), layer_norm=c.layer_norm,
                                      drop_rate=c.drop_rate)
        for t in ag.target_q_nets:
            for p in t.parameters():
                p.requires_grad = False
        ag.q_optim = th.optim.Adam([p for net in ag.q_nets for p in net.parameters()], lr=c.lr)
    for n in range(2):
        _load(ag.q_nets[n], inp["q"][n])
        _load(ag.target_q_nets[n], inp["tq"][n])
    _load(ag.policy, inp["pol"])
    _load(ag.target_policy, inp["tpol"])
    qp = [p for net in ag.q_nets for p in net.parameters()]
    _seed_opt(ag.q_optim, qp, inp["q_state"], c.step)
    _seed_opt(ag.policy_optim, list(ag.policy.parameters()), inp["p_state"], c.step)
    batch = tuple(_t(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones"))
    if c.per:
        batch = batch + (th.arange(c.B),)
    ag._sample_batch_experiences = lambda: batch
    rec = {}
    ag.replay_buffer.update_priorities = lambda idx, pr: rec.__setitem__("priority", np.asarray(pr, np.float64).copy())
    ag.weight_support = [_t(s) for s in inp["support"]] if c.n_support > 1 else [_t(inp["support"][0])]
    ag._n_updates = c.n_updates
    do_policy = c.n_updates % 2 == 0
    ag.log, ag.global_step = do_policy, 100
    drops = []
    if c.drop_rate > 0:
        for key in ("target", "q") + (("q_pi",) if do_policy else ()):
            for n in range(2):
                drops += inp["drop"][key][n]
    with Pins(randn_like=[inp["eps_next"]], dropout=drops, choices=list(inp["choice"])) as pins:
        ag.update(_t(inp["weight"]))
    out = {}
    if do_policy:
        out["critic_loss"] = np.float32(pins.logged["losses/critic_loss"])
        out["policy_loss"] = np.float32(pins.logged["losses/policy_loss"])
    if c.per:
        out["priority"] = rec["priority"]
    for n in range(2):
        _dump(out, f"q{n}", list(ag.q_nets[n].parameters()), ag.q_optim, c.subsample)
        _dump(out, f"tq{n}", list(ag.target_q_nets[n].parameters()), None, c.subsample)
    _dump(out, "pol", list(ag.policy.parameters()), ag.policy_optim if do_policy else None, c.subsample)
    _dump(out, "tpol", list(ag.target_policy.parameters()), None, c.subsample)
    return out


def main():
    ref = rh.import_reference_ac()
    th.set_num_threads(1)
    runners = dict(capql=run_capql, mosac=run_mosac, gpipd=run_gpipd, sacd=run_sacd)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for c in AC_CASES:
        if only and c.name not in only and c.algo not in only:
            continue
        out = runners[c.algo](ref, c)
        np.savez_compressed(os.path.join(HERE, f"ac_{c.name}.npz"), **out)
        print(c.name, {k: float(v) for k, v in out.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
