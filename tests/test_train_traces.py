"""End-to-end training-loop parity on the CPU test backend: the HIP agents (kernel sources under the wave emulator) run
the SAME short training as the unmodified reference did for tests/golden/train_traces_ac.npz -- same environment, same
seeds, same initial parameters -- and must take the same actions and end with the same parameters.  This pins the host
loops (RNG consumption order of torch / numpy / random / the environment, buffer semantics, schedules, delayed and
target updates) on top of the per-update parity of the kernel tests."""
import os

import numpy as np
import pytest
import torch as th

import momdp
import train_cases as tc

import morl_baselines_amd.native as native

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_traces_ac.npz")


class _Backend:
    """What the trace tests need: the library, the device string the agents get, and the RNG placement."""

    def __init__(self, lib, device):
        self.lib, self.device = lib, device


@pytest.fixture(scope="module", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def sim(request):
    """CPU: the kernel sources under the wave emulator.  MI355X: the real library, with every noise draw taken from torch's
    CPU generator (acnets.HOST_NOISE) -- the stream the reference consumed when the golden traces were recorded -- so the
    GPU run replays the reference's seeded run draw for draw."""
    import morl_baselines_amd.acnets as acnets
    if request.param == "sim":
        import simlib
        lib = simlib.load_sim()
        native.use_library(lib)
        yield _Backend(lib, "cpu")
        native.use_library(None)
        return
    acnets.HOST_NOISE = True
    yield _Backend(native.load_library(), "cuda:0")
    acnets.HOST_NOISE = False


def load_init(g, prefix, modules):
    params = [p for m in modules for p in m.parameters()]
    with th.no_grad():
        for i, p in enumerate(params):
            p.copy_(th.tensor(g[f"{prefix}_{i}"]))
    return params


def check_final(g, prefix, params, atol):
    worst = 0.0
    for i, p in enumerate(params):
        want = g[f"{prefix}_{i}"]
        err = float(np.abs(p.detach().cpu().numpy() - want).max())
        worst = max(worst, err)
        assert err <= atol, f"{prefix}_{i}: max |diff| {err:.3e}"
    return worst


def test_capql_trace(sim):
    from morl_baselines_amd.capql import CAPQL
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED)
    ag = CAPQL(env, log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, **tc.CAPQL)
    params = load_init(g, "capql_init", ag.q_nets + ag.target_q_nets + [ag.policy])
    tc.reseed()
    ag.train(total_timesteps=tc.CAPQL_STEPS)
    np.testing.assert_allclose(np.asarray(env.action_log), g["capql_actions"], rtol=0, atol=2e-5)
    worst = check_final(g, "capql_final", params, atol=3e-5)
    print(f"\nCAPQL: {tc.CAPQL_STEPS} steps / {ag._q_step} updates, max parameter deviation {worst:.2e}")


def test_mosac_trace(sim):
    from morl_baselines_amd.mosac import MOSAC
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED)
    ag = MOSAC(env, tc.MOSAC_WEIGHTS.copy(), log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, **tc.MOSAC)
    params = load_init(g, "mosac_init", [ag.actor, ag.qf1, ag.qf2, ag.qf1_target, ag.qf2_target])
    tc.reseed()
    ag.train(total_timesteps=tc.MOSAC_STEPS)
    np.testing.assert_allclose(np.asarray(env.action_log), g["mosac_actions"], rtol=0, atol=5e-5)
    worst = check_final(g, "mosac_final", params, atol=1e-4)
    np.testing.assert_allclose(ag.log_alpha.cpu().numpy(), g["mosac_log_alpha"], rtol=0, atol=2e-5)
    print(f"\nMOSAC: {tc.MOSAC_STEPS} steps / {ag._q_step} updates, max parameter deviation {worst:.2e}")


def test_gpils_continuous_trace(sim):
    from morl_baselines_amd.gpi_pd_continuous import GPILSContinuousAction
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED)
    ag = GPILSContinuousAction(env, log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, q_drop_rate=0.0, **tc.GPILS_CONT)
    params = load_init(g, "gpic_init", ag.q_nets + ag.target_q_nets + [ag.policy, ag.target_policy])
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPILS_CONT_STEPS, weight=tc.WEIGHT.copy(),
                       weight_support=[s.copy() for s in tc.SUPPORT], change_weight_every_episode=True)
    np.testing.assert_allclose(np.asarray(env.action_log), g["gpic_actions"], rtol=0, atol=5e-5)
    worst = check_final(g, "gpic_final", params, atol=1e-4)
    assert float(ag.replay_buffer.tree.nodes[0][0]) == pytest.approx(float(g["gpic_tree_root"]), rel=1e-4)
    print(f"\nGPI-LS continuous: {tc.GPILS_CONT_STEPS} steps / {ag._n_updates} updates, max parameter deviation {worst:.2e}")


def test_gpils_discrete_trace(sim):
    from morl_baselines_amd.gpi_pd import GPILS
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED)
    ag = GPILS(env, log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, **tc.GPILS)
    params = load_init(g, "gpi_init", ag.q_nets + ag.target_q_nets)
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPILS_STEPS, weight=tc.WEIGHT.copy(), weight_support=[s.copy() for s in tc.SUPPORT],
                       change_w_every_episode=True)
    assert np.array_equal(np.asarray(env.action_log, dtype=np.int8), g["gpi_actions"])
    worst = check_final(g, "gpi_final", params, atol=1e-4)
    assert float(ag.replay_buffer.tree.nodes[0][0]) == pytest.approx(float(g["gpi_tree_root"]), rel=1e-4)
    print(f"\nGPI-LS: {tc.GPILS_STEPS} steps / {ag._adam_step} updates, max parameter deviation {worst:.2e}")


def test_envelope_trace(sim):
    from morl_baselines_amd.envelope import Envelope
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED)
    ag = Envelope(env, log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, **tc.ENVELOPE)
    params = load_init(g, "env_init", [ag.q_net, ag.target_q_net])
    tc.reseed()
    ag.train(total_timesteps=tc.ENVELOPE_STEPS)
    assert np.array_equal(np.asarray(env.action_log, dtype=np.int8), g["env_actions"])
    worst = check_final(g, "env_final", params, atol=1e-4)
    assert float(ag.replay_buffer.tree.nodes[0][0]) == pytest.approx(float(g["env_tree_root"]), rel=1e-4)
    np.testing.assert_allclose([ag.epsilon, ag.homotopy_lambda], g["env_eps_lambda"], rtol=1e-12)
    print(f"\nEnvelope: {tc.ENVELOPE_STEPS} steps / {ag._adam_step} updates, max parameter deviation {worst:.2e}")


def test_mosac_discrete_trace(sim):
    from morl_baselines_amd.mosac_discrete import MOSACDiscrete
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED)
    ag = MOSACDiscrete(env, tc.SACD_WEIGHTS.copy(), log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, **tc.SACD)
    params = load_init(g, "sacd_init", [ag.actor, ag.qf1, ag.qf2, ag.qf1_target, ag.qf2_target])
    tc.reseed()
    ag.train(total_timesteps=tc.SACD_STEPS)
    assert np.array_equal(np.asarray(env.action_log, dtype=np.int8), g["sacd_actions"])
    worst = check_final(g, "sacd_final", params, atol=1e-4)
    np.testing.assert_allclose(ag.log_alpha.cpu().numpy(), g["sacd_log_alpha"], rtol=0, atol=2e-5)
    print(f"\nMOSAC discrete: {tc.SACD_STEPS} steps / {ag._q_step} updates, max parameter deviation {worst:.2e}")


def test_gpipd_dyna_trace(sim):
    """GPI-PD proper: dynamics-ensemble fits (early stopping, elites), imagined rollouts with batched GPI actions and
    model noise, real / imagined batch mixing, PER + GPI priorities -- the whole loop against the reference's seeded run."""
    from morl_baselines_amd.gpi_pd import GPIPD
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.TreasureLine(tc.SEED, env_id=tc.GPIPD_DYNA_ENV_ID)
    ag = GPIPD(env, log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, dynamics_train_freq=lambda t: tc.GPIPD_DYNA_TRAIN_FREQ,
               dynamics_max_rows=256, **tc.GPIPD_DYNA)
    params = load_init(g, "dyna_init", ag.q_nets + ag.target_q_nets)
    ag.dynamics_fit_kwargs = dict(tc.GPIPD_DYNA_FIT)
    nl = len(tc.GPIPD_DYNA["dynamics_net_arch"]) + 1
    sd = ag.dynamics.state_dict()
    ag.dynamics.load_state_dict({**sd, **{f"layers.{l}.W": th.tensor(g[f"dyna_model_init_W{l}"]) for l in range(nl)},
                                 **{f"layers.{l}.b": th.tensor(g[f"dyna_model_init_b{l}"]) for l in range(nl)}})
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPIPD_DYNA_STEPS, weight=tc.WEIGHT.copy(), weight_support=[s.copy() for s in tc.SUPPORT],
                       change_w_every_episode=True)
    assert np.array_equal(np.asarray(env.action_log, dtype=np.int8), g["dyna_actions"])
    assert [len(ag.dynamics_buffer), ag.dynamics_buffer.ptr] == g["dyna_model_buffer"].tolist()
    nb = len(ag.dynamics_buffer)
    np.testing.assert_allclose(ag.dynamics_buffer.obs[:nb], g["dyna_model_obs"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(ag.dynamics_buffer.rewards[:nb], g["dyna_model_rewards"], rtol=0, atol=2e-4)
    sd = ag.dynamics.state_dict()
    for l in range(nl):
        np.testing.assert_allclose(sd[f"layers.{l}.W"].cpu().numpy(), g[f"dyna_model_W{l}"], rtol=0, atol=1e-4)
    worst = check_final(g, "dyna_final", params, atol=2e-4)
    assert float(ag.replay_buffer.tree.nodes[0][0]) == pytest.approx(float(g["dyna_tree_root"]), rel=1e-3)
    print(f"\nGPI-PD + Dyna: {tc.GPIPD_DYNA_STEPS} steps / {ag._adam_step} updates, {nb} imagined transitions, "
          f"max parameter deviation {worst:.2e}")


def test_gpipd_continuous_dyna_trace(sim):
    """GPI-PD with continuous actions and the Dyna model: ensemble fits, roll-outs with the noisy TD3 policy, the
    uncertainty filter, mixed real / imagined batches with PER on the real half -- against the reference's seeded run."""
    from morl_baselines_amd.gpi_pd_continuous import GPIPDContinuousAction
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env = momdp.PointReach(tc.SEED, env_id=tc.GPIPD_CONT_DYNA_ENV_ID)
    ag = GPIPDContinuousAction(env, log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, q_drop_rate=0.0, dynamics_max_rows=256,
                               **tc.GPIPD_CONT_DYNA)
    params = load_init(g, "dynac_init", ag.q_nets + ag.target_q_nets + [ag.policy, ag.target_policy])
    ag.dynamics_fit_kwargs = dict(tc.GPIPD_DYNA_FIT)
    nl = len(tc.GPIPD_CONT_DYNA["dynamics_net_arch"]) + 1
    sd = ag.dynamics.state_dict()
    ag.dynamics.load_state_dict({**sd, **{f"layers.{l}.W": th.tensor(g[f"dynac_model_init_W{l}"]) for l in range(nl)},
                                 **{f"layers.{l}.b": th.tensor(g[f"dynac_model_init_b{l}"]) for l in range(nl)}})
    tc.reseed()
    ag.train_iteration(total_timesteps=tc.GPIPD_CONT_DYNA_STEPS, weight=tc.WEIGHT.copy(),
                       weight_support=[s.copy() for s in tc.SUPPORT], change_weight_every_episode=True)
    np.testing.assert_allclose(np.asarray(env.action_log), g["dynac_actions"], rtol=0, atol=5e-5)
    assert [len(ag.dynamics_buffer), ag.dynamics_buffer.ptr] == g["dynac_model_buffer"].tolist()
    nb = len(ag.dynamics_buffer)
    np.testing.assert_allclose(ag.dynamics_buffer.obs[:nb], g["dynac_model_obs"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(ag.dynamics_buffer.actions[:nb], g["dynac_model_actions"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(ag.dynamics_buffer.rewards[:nb], g["dynac_model_rewards"], rtol=0, atol=2e-4)
    worst = check_final(g, "dynac_final", params, atol=2e-4)
    assert float(ag.replay_buffer.tree.nodes[0][0]) == pytest.approx(float(g["dynac_tree_root"]), rel=1e-3)
    print(f"\nGPI-PD continuous + Dyna: {tc.GPIPD_CONT_DYNA_STEPS} steps / {ag._n_updates} updates, {nb} imagined transitions, "
          f"max parameter deviation {worst:.2e}")


def test_morld_trace(sim):
    """MORL/D: turn-by-turn candidate training on the shared environment and buffer, batched updates of the other
    sub-problem learners, stochastic evaluations into the Pareto archive, PSA weight adaptation -- against the reference's
    seeded run (pymoo's weight initialisation replaced by fixed weights in both)."""
    from morl_baselines_amd.morld import MORLD
    g = np.load(GOLD)
    tc.reseed(tc.SEED)
    env, eval_env = momdp.PointReach(tc.SEED), momdp.PointReach(tc.SEED + 1)
    ag = MORLD(env, log=False, seed=tc.SEED, device=sim.device, lib=sim.lib, weights=tc.MORLD_WEIGHTS.copy(), **tc.MORLD)
    params = []
    for k, pol in enumerate(ag.population):
        w = pol.wrapped
        params.append(load_init(g, f"morld_init_{k}", [w.actor, w.qf1, w.qf2, w.qf1_target, w.qf2_target]))
    tc.reseed()
    ag.train(total_timesteps=tc.MORLD_STEPS, eval_env=eval_env, ref_point=np.zeros(2), num_eval_episodes_for_front=1,
             checkpoints=False)
    np.testing.assert_allclose(np.asarray(env.action_log), g["morld_actions"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(np.asarray(eval_env.action_log), g["morld_eval_actions"], rtol=0, atol=2e-4)
    worst = 0.0
    for k, pol in enumerate(ag.population):
        worst = max(worst, check_final(g, f"morld_final_{k}", params[k], atol=2e-4))
        np.testing.assert_allclose(pol.wrapped.log_alpha.cpu().numpy(), g[f"morld_log_alpha_{k}"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(np.stack([p.weights for p in ag.population]), g["morld_weights"], rtol=1e-12)
    np.testing.assert_allclose(np.stack(ag.archive.evaluations), g["morld_archive"], rtol=1e-4, atol=1e-4)
    print(f"\nMORL/D: {tc.MORLD_STEPS} steps, 3 learners, {len(ag.archive.evaluations)} archive entries, "
          f"max parameter deviation {worst:.2e}")
