"""End-to-end training traces of the UNMODIFIED reference agents (CAPQL, MOSAC, GPI-LS continuous, GPI-LS) on the
self-contained MOMDPs of tests/momdp.py, CPU, a few dozen environment steps with learning.  The HIP agents are run on
the same environments with the same seeds on the CPU test backend (tests/test_train_traces.py): they must take the same
actions and end with the same parameters -- i.e. the host loops consume every RNG stream (torch, numpy, random, the
environment's) exactly as the reference does and the update arithmetic agrees step after step.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_train_ac.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import ref_harness as rh  # noqa: E402
import train_cases as tc  # noqa: E402


def params_of(mods):
    return [p.detach().numpy().copy() for m in mods for p in m.parameters()]


def dump(out, prefix, arrs):
    for i, a in enumerate(arrs):
        out[f"{prefix}_{i}"] = a


def main():
    ref = rh.import_reference_ac()
    from morl_baselines.multi_policy.gpi_pd import gpi_pd as gpi_mod
    import momdp

    th.set_num_threads(1)
    noop = lambda *a, **k: None  # noqa: E731
    out = {}

    # ---- CAPQL -------------------------------------------------------------------------------------------------------
    ref.capql.equally_spaced_weights = noop
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED)
    ag = ref.capql.CAPQL(env, log=False, seed=tc.SEED, device="cpu", **tc.CAPQL)
    nets = ag.q_nets + ag.target_q_nets + [ag.policy]
    dump(out, "capql_init", params_of(nets))
    tc.reseed()
    ag.train(total_timesteps=tc.CAPQL_STEPS, eval_env=None, ref_point=np.zeros(2))
    dump(out, "capql_final", params_of(nets))
    out["capql_actions"] = np.asarray(env.action_log)

    # ---- MOSAC -------------------------------------------------------------------------------------------------------
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED)
    ag = ref.mosac.MOSAC(env, weights=tc.MOSAC_WEIGHTS.copy(), log=False, seed=tc.SEED, device="cpu", **tc.MOSAC)
    nets = [ag.actor, ag.qf1, ag.qf2, ag.qf1_target, ag.qf2_target]
    dump(out, "mosac_init", params_of(nets))
    tc.reseed()
    ag.train(total_timesteps=tc.MOSAC_STEPS)
    dump(out, "mosac_final", params_of(nets))
    out["mosac_actions"] = np.asarray(env.action_log)
    out["mosac_log_alpha"] = ag.log_alpha.detach().numpy().copy()

    # ---- GPI-LS continuous (critics without dropout: the device generator cannot replay torch's bernoulli draws) ------
    mod = ref.gpipd_cont
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED)
    ag = mod.GPILSContinuousAction(env, log=False, seed=tc.SEED, device="cpu", **tc.GPILS_CONT)
    for lst in (ag.q_nets, ag.target_q_nets):
        for n in range(2):
            lst[n] = mod.QNetwork(2, 1, 2, net_arch=tc.GPILS_CONT["net_arch"], layer_norm=True, drop_rate=0.0)
    for q, t in zip(ag.q_nets, ag.target_q_nets):
        t.load_state_dict(q.state_dict())
        for p in t.parameters():
            p.requires_grad = False
    ag.q_optim = th.optim.Adam([p for net in ag.q_nets for p in net.parameters()], lr=ag.learning_rate)
    nets = ag.q_nets + ag.target_q_nets + [ag.policy, ag.target_policy]
    dump(out, "gpic_init", params_of(nets))
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPILS_CONT_STEPS, weight=tc.WEIGHT.copy(), weight_support=[s.copy() for s in tc.SUPPORT],
                       change_weight_every_episode=True)
    dump(out, "gpic_final", params_of(nets))
    out["gpic_actions"] = np.asarray(env.action_log)
    out["gpic_tree_root"] = np.float64(ag.replay_buffer.tree.nodes[0][0])

    # ---- GPI-LS (discrete) ----------------------------------------------------------------------------------------------
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED)
    ag = gpi_mod.GPILS(env, log=False, seed=tc.SEED, device="cpu", **tc.GPILS)
    nets = ag.q_nets + ag.target_q_nets
    dump(out, "gpi_init", params_of(nets))
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPILS_STEPS, weight=tc.WEIGHT.copy(), weight_support=[s.copy() for s in tc.SUPPORT],
                       change_w_every_episode=True)
    dump(out, "gpi_final", params_of(nets))
    out["gpi_actions"] = np.asarray(env.action_log, dtype=np.int8)
    out["gpi_tree_root"] = np.float64(ag.replay_buffer.tree.nodes[0][0])

    # ---- GPI-PD with the Dyna model (dynamics ensemble fit, imagined rollouts, mixed batches) -----------------------------------
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED, env_id=tc.GPIPD_DYNA_ENV_ID)
    ag = gpi_mod.GPIPD(env, log=False, seed=tc.SEED, device="cpu",
                       dynamics_train_freq=lambda t: tc.GPIPD_DYNA_TRAIN_FREQ, **tc.GPIPD_DYNA)
    nets = ag.q_nets + ag.target_q_nets
    ref_fit = ag.dynamics.fit
    ag.dynamics.fit = lambda X, Y: ref_fit(X, Y, **tc.GPIPD_DYNA_FIT)      # instance attribute: same fit(), fewer epochs
    dump(out, "dyna_init", params_of(nets))
    for l, layer in enumerate(ag.dynamics.layers):
        out[f"dyna_model_init_W{l}"], out[f"dyna_model_init_b{l}"] = layer.W.detach().numpy().copy(), layer.b.detach().numpy().copy()
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPIPD_DYNA_STEPS, weight=tc.WEIGHT.copy(), weight_support=[s.copy() for s in tc.SUPPORT],
                       change_w_every_episode=True)
    dump(out, "dyna_final", params_of(nets))
    for l, layer in enumerate(ag.dynamics.layers):
        out[f"dyna_model_W{l}"], out[f"dyna_model_b{l}"] = layer.W.detach().numpy().copy(), layer.b.detach().numpy().copy()
    out["dyna_actions"] = np.asarray(env.action_log, dtype=np.int8)
    out["dyna_model_buffer"] = np.array([len(ag.dynamics_buffer), ag.dynamics_buffer.ptr])
    nb = len(ag.dynamics_buffer)
    out["dyna_model_obs"] = ag.dynamics_buffer.obs[:nb].copy()
    out["dyna_model_rewards"] = ag.dynamics_buffer.rewards[:nb].copy()
    out["dyna_tree_root"] = np.float64(ag.replay_buffer.tree.nodes[0][0])
    print("dyna: model buffer", nb, "elites", ag.dynamics.elites)

    # ---- GPI-PD continuous with the Dyna model (critics without dropout, as above) ---------------------------------------------
    mod = ref.gpipd_cont
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED, env_id=tc.GPIPD_CONT_DYNA_ENV_ID)
    ag = mod.GPIPDContinuousAction(env, log=False, seed=tc.SEED, device="cpu", **tc.GPIPD_CONT_DYNA)
    for lst in (ag.q_nets, ag.target_q_nets):
        for n in range(2):
            lst[n] = mod.QNetwork(2, 1, 2, net_arch=tc.GPIPD_CONT_DYNA["net_arch"], layer_norm=True, drop_rate=0.0)
    for q, t in zip(ag.q_nets, ag.target_q_nets):
        t.load_state_dict(q.state_dict())
        for p in t.parameters():
            p.requires_grad = False
    ag.q_optim = th.optim.Adam([p for net in ag.q_nets for p in net.parameters()], lr=ag.learning_rate)
    ref_fit_c = ag.dynamics.fit
    ag.dynamics.fit = lambda X, Y: ref_fit_c(X, Y, **tc.GPIPD_DYNA_FIT)
    uncs, ref_step = [], mod.ModelEnv.step

    def recording_step(self, obs, act, deterministic=False):     # observe only: which uncertainties does the filter see?
        res = ref_step(self, obs, act, deterministic)
        uncs.append(np.asarray(res[3]["uncertainty"]).copy())
        return res
    mod.ModelEnv.step = recording_step
    nets = ag.q_nets + ag.target_q_nets + [ag.policy, ag.target_policy]
    dump(out, "dynac_init", params_of(nets))
    for l, layer in enumerate(ag.dynamics.layers):
        out[f"dynac_model_init_W{l}"], out[f"dynac_model_init_b{l}"] = layer.W.detach().numpy().copy(), layer.b.detach().numpy().copy()
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPIPD_CONT_DYNA_STEPS, weight=tc.WEIGHT.copy(),
                       weight_support=[s.copy() for s in tc.SUPPORT], change_weight_every_episode=True)
    dump(out, "dynac_final", params_of(nets))
    out["dynac_actions"] = np.asarray(env.action_log)
    nb = len(ag.dynamics_buffer)
    out["dynac_model_buffer"] = np.array([nb, ag.dynamics_buffer.ptr])
    out["dynac_model_obs"] = ag.dynamics_buffer.obs[:nb].copy()
    out["dynac_model_actions"] = ag.dynamics_buffer.actions[:nb].copy()
    out["dynac_model_rewards"] = ag.dynamics_buffer.rewards[:nb].copy()
    out["dynac_tree_root"] = np.float64(ag.replay_buffer.tree.nodes[0][0])
    mod.ModelEnv.step = ref_step
    print("dyna continuous: model buffer", nb, "elites", ag.dynamics.elites, "uncertainty quantiles",
          np.quantile(np.concatenate(uncs), [0.0, 0.25, 0.5, 0.75, 1.0]))

    # ---- MORL/D: population of MOSAC learners, shared buffer, transfer, PSA weight adaptation ---------------------------------
    from morl_baselines.multi_policy.morld import morld as morld_mod
    morld_mod.equally_spaced_weights = lambda dim, n, seed=None: [w.copy() for w in tc.MORLD_WEIGHTS]   # pymoo is absent
    tc.reseed(tc.SEED)
    env, eval_env = momdp.PointReach(tc.SEED), momdp.PointReach(tc.SEED + 1)
    ag = morld_mod.MORLD(env, log=False, seed=tc.SEED, device="cpu", **tc.MORLD)
    for k, pol in enumerate(ag.population):
        w = pol.wrapped
        dump(out, f"morld_init_{k}", params_of([w.actor, w.qf1, w.qf2, w.qf1_target, w.qf2_target]))
    tc.reseed()
    ag.train(total_timesteps=tc.MORLD_STEPS, eval_env=eval_env, ref_point=np.zeros(2), num_eval_episodes_for_front=1,
             checkpoints=False)
    for k, pol in enumerate(ag.population):
        w = pol.wrapped
        dump(out, f"morld_final_{k}", params_of([w.actor, w.qf1, w.qf2, w.qf1_target, w.qf2_target]))
        out[f"morld_log_alpha_{k}"] = w.log_alpha.detach().numpy().copy()
    out["morld_actions"] = np.asarray(env.action_log)
    out["morld_eval_actions"] = np.asarray(eval_env.action_log)
    out["morld_weights"] = np.stack([np.asarray(p_.weights, dtype=np.float64) for p_ in ag.population])
    out["morld_archive"] = np.stack(ag.archive.evaluations)
    print("morld: archive", len(ag.archive.evaluations), "weights", out["morld_weights"].round(3).tolist())

    # ---- MOSAC with discrete actions -------------------------------------------------------------------------------------------
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED)
    ag = ref.sacd.MOSACDiscrete(env, weights=tc.SACD_WEIGHTS.copy(), log=False, seed=tc.SEED, device="cpu", **tc.SACD)
    nets = [ag.actor, ag.qf1, ag.qf2, ag.qf1_target, ag.qf2_target]
    dump(out, "sacd_init", params_of(nets))
    tc.reseed()
    ag.train(total_timesteps=tc.SACD_STEPS)
    dump(out, "sacd_final", params_of(nets))
    out["sacd_actions"] = np.asarray(env.action_log, dtype=np.int8)
    out["sacd_log_alpha"] = ag.log_alpha.detach().numpy().copy()

    # ---- Envelope (epsilon / homotopy schedules, PER, periodic target copy) --------------------------------------------------
    refe = rh.import_reference()
    refe.envelope.equally_spaced_weights = noop
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED)
    ag = refe.envelope.Envelope(env, log=False, seed=tc.SEED, device="cpu", **tc.ENVELOPE)
    nets = [ag.q_net, ag.target_q_net]
    dump(out, "env_init", params_of(nets))
    tc.reseed()
    ag.train(total_timesteps=tc.ENVELOPE_STEPS)
    dump(out, "env_final", params_of(nets))
    out["env_actions"] = np.asarray(env.action_log, dtype=np.int8)
    out["env_tree_root"] = np.float64(ag.replay_buffer.tree.nodes[0][0])
    out["env_eps_lambda"] = np.array([ag.epsilon, ag.homotopy_lambda])

    np.savez_compressed(os.path.join(HERE, "train_traces_ac.npz"), **out)
    for k in ("capql_actions", "mosac_actions", "gpic_actions", "gpi_actions", "dyna_actions", "sacd_actions", "env_actions"):
        print(k, out[k].shape, np.asarray(out[k]).reshape(len(out[k]), -1)[-3:].tolist())


if __name__ == "__main__":
    main()
