"""Host-side mirrors of the reference's continuous-action agents (CAPQL, MOSAC, MORL/D, GPI-LS continuous) on the HIP
engine: replay semantics, one ``update()`` against the oracle on the very batch / noise the agent drew, checkpoint
round trips and short training runs.  ``sim`` = kernel sources under the host wave emulator, ``hip`` = gfx950 (-m gpu)."""
import os
import random

import numpy as np
import pytest
import torch as th

import ac_oracle as ac
import momdp

import morl_baselines_amd.native as native
from morl_baselines_amd.capql import CAPQL, ReplayMemory, WeightSamplerAngle
from morl_baselines_amd.gpi_pd_continuous import GPILSContinuousAction, GPIPDContinuousAction
from morl_baselines_amd.morld import MORLD, simplex_lattice_weights
from morl_baselines_amd.mosac import MOSAC


class BoxEnv(momdp.PointReach):
    """PointReach with a wider observation / action vector so the network shapes are less degenerate."""

    def __init__(self, seed=0, D=5, Ad=2, low=-1.0, high=1.0):
        super().__init__(seed)
        self.observation_space = momdp.BoxSpace(-1.0, 1.0, (D,), seed)
        self.action_space = momdp.BoxSpace(low, high, (Ad,), seed)
        self._D = D

    def _obs(self):
        o = np.zeros(self._D, dtype=np.float32)
        o[0], o[1] = self.x, self.t / self.HORIZON
        o[2:] = np.sin(np.arange(2, self._D) * (1.0 + self.x))
        return o


@pytest.fixture(scope="module", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    if request.param == "sim":
        import simlib
        lib = simlib.load_sim()
        native.use_library(lib)
        yield lib, th.device("cpu")
        native.use_library(None)
        return
    yield native.load_library(), th.device("cuda:0")


def rng_snapshot(dev):
    return (random.getstate(), np.random.get_state(), th.get_rng_state(),
            th.cuda.get_rng_state(dev) if dev.type == "cuda" else None)


def rng_restore(snap, dev):
    random.setstate(snap[0])
    np.random.set_state(snap[1])
    th.set_rng_state(snap[2])
    if snap[3] is not None:
        th.cuda.set_rng_state(snap[3], dev)


def fill_capql(ag, env, n, seed=0):
    rng = np.random.default_rng(seed)
    D, Ad, R = ag.observation_dim, ag.action_dim, ag.reward_dim
    for _ in range(n):
        w = np.abs(rng.standard_normal(R)); w /= w.sum()
        ag.replay_buffer.push(rng.standard_normal(D).astype(np.float32), rng.uniform(-1, 1, Ad).astype(np.float32),
                              w.astype(np.float32), rng.standard_normal(R).astype(np.float32),
                              rng.standard_normal(D).astype(np.float32), rng.random() < 0.1)


def fill_buffer(buf, n, D, Ad, R, seed=0, low=-1.0, high=1.0):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        buf.add(rng.standard_normal(D).astype(np.float32), rng.uniform(low, high, Ad).astype(np.float32),
                rng.standard_normal(R).astype(np.float32), rng.standard_normal(D).astype(np.float32), rng.random() < 0.1)


def cpu(vs):
    return [v.detach().cpu().clone() for v in vs]


def assert_lists_close(got, want, rtol=2e-4, frac=2e-5):
    for g, w in zip(got, want):
        scale = float(w.abs().max()) + 1e-30
        np.testing.assert_allclose(g.cpu().numpy(), w.numpy(), rtol=rtol, atol=frac * scale)


# ---------------------------------------------------------------------------------------------------------------------
def test_replay_memory_matches_reference_sampling(be):
    lib, dev = be
    mem = ReplayMemory(40, device=dev, lib=lib)
    rng = np.random.default_rng(0)
    items = []
    for t in range(55):                                  # wraps the ring
        it = (rng.standard_normal(4), rng.standard_normal(2), rng.random(3), rng.standard_normal(3),
              rng.standard_normal(4), t % 7 == 0)
        items.append(it)
        mem.push(*it)
    assert len(mem) == 40 and mem.position == 55 % 40
    random.seed(3)
    got = mem.sample(16, to_tensor=True)
    random.seed(3)
    want = random.sample(mem.buffer, 16)                # capql.py:58
    for k in range(5):
        np.testing.assert_array_equal(got[k].cpu().numpy(), np.stack([w[k] for w in want]).astype(np.float32))
    np.testing.assert_array_equal(got[5].cpu().numpy(), np.stack([w[5] for w in want]).astype(np.float32))
    random.seed(4)
    host = mem.sample(8, to_tensor=False)
    random.seed(4)
    want = random.sample(mem.buffer, 8)
    np.testing.assert_array_equal(host[0], np.stack([w[0] for w in want]))


def test_weight_sampler_angle_properties():
    th.manual_seed(0)
    ws = WeightSamplerAngle(3, th.pi * 22.5 / 180).sample(200)
    assert ws.shape == (200, 3)
    np.testing.assert_allclose(ws.abs().sum(1).numpy(), 1.0, rtol=1e-5)
    centre = th.ones(3) / th.norm(th.ones(3))
    cosang = (ws / th.norm(ws, dim=1, keepdim=True)) @ centre
    assert (th.acos(cosang.clamp(-1, 1)) <= th.pi * 22.5 / 180 + 1e-4).all()


def test_gradient_updates_loop_in_one_call_equals_sequential_updates(be):
    """``CAPQL.update`` with ``gradient_updates = 3`` runs its loop as ONE library entry (``morl_ac_update_n``: batches and
    noise drawn beforehand in the reference's RNG order); it must take exactly the steps of three single-update calls."""
    lib, dev = be
    runs = []
    for gu, calls in ((3, 1), (1, 3)):
        env = BoxEnv(low=-2.0, high=1.0)
        th.manual_seed(0)
        ag = CAPQL(env, net_arch=[32, 32], batch_size=16, buffer_size=256, learning_starts=10, log=False, seed=0,
                   device=dev, lib=lib, alpha=0.1, gradient_updates=gu)
        fill_capql(ag, env, 80)
        random.seed(11)
        th.manual_seed(11)
        for _ in range(calls):
            ag.update()
        e = ag.engine
        runs.append((e.q.clone().cpu(), e.pol.clone().cpu(), e.q_target.clone().cpu(), ag.last_losses(), ag._q_step))
    a, b = runs
    assert th.equal(a[0], b[0]) and th.equal(a[1], b[1]) and th.equal(a[2], b[2]) and a[3] == b[3] and a[4] == b[4] == 3


def test_capql_update_matches_oracle_and_checkpoints(be, tmp_path):
    lib, dev = be
    env = BoxEnv(low=-2.0, high=1.0)
    th.manual_seed(0)
    ag = CAPQL(env, net_arch=[32, 32], batch_size=16, buffer_size=256, learning_starts=10, log=False, seed=0,
               device=dev, lib=lib, alpha=0.1)
    fill_capql(ag, env, 80)
    e = ag.engine
    q0 = [cpu(e.q_views(e.q, 0, n)) for n in range(2)]
    tq0 = [cpu(e.q_views(e.q_target, 0, n)) for n in range(2)]
    pol0 = cpu(e.policy_views(e.pol))
    random.seed(7)
    th.manual_seed(7)
    snap = rng_snapshot(dev)
    ag.update()
    rng_restore(snap, dev)
    batch = ag._sample_batch_experiences()
    eps = [th.randn((16, ag.action_dim), dtype=th.float32, device=dev) for _ in range(2)]
    qspec = ac.MlpSpec(env.observation_space.shape[0] + 2 + 2, (32, 32), 2)
    trunk = ac.MlpSpec(env.observation_space.shape[0] + 2, (32, 32))
    qs, ps = (dict(exp_avg=ac.zeros_like(sum(q0, [])), exp_avg_sq=ac.zeros_like(sum(q0, []))),
              dict(exp_avg=ac.zeros_like(pol0), exp_avg_sq=ac.zeros_like(pol0)))
    out = ac.capql_update(qspec, trunk, q0, tq0, pol0, qs, ps, tuple(b.cpu() for b in batch), eps[0].cpu(),
                          eps[1].cpu(), e.action_scale.cpu(), e.action_bias.cpu(), gamma=ag.gamma, alpha=0.1,
                          lr=ag.learning_rate, tau=ag.tau, step=1)
    closs, ploss = ag.last_losses()
    assert abs(closs - float(out["critic_loss"])) <= 1e-5 * abs(float(out["critic_loss"]))
    assert abs(ploss - float(out["policy_loss"])) <= 1e-5 * max(abs(float(out["policy_loss"])), 1e-3)
    assert_lists_close(e.policy_views(e.pol_exp_avg), ps["exp_avg"])
    for n in range(2):
        assert_lists_close(e.q_views(e.q_exp_avg, 0, n), qs["exp_avg"][n * 6:(n + 1) * 6])
        assert_lists_close(e.q_views(e.q_target, 0, n), tq0[n], rtol=1e-5, frac=1e-6)
    # deterministic action = tanh(mean) * scale + bias of the (updated) policy
    obs, w = np.linspace(-1, 1, 5).astype(np.float32), np.array([0.3, 0.7], dtype=np.float32)
    want = ac.capql_policy_action(trunk, cpu(e.policy_views(e.pol)), th.tensor(obs)[None], th.tensor(w)[None],
                                  e.action_scale.cpu(), e.action_bias.cpu())[0]
    np.testing.assert_allclose(ag.eval(obs, w), want.numpy(), rtol=1e-5, atol=1e-6)
    # checkpoint round trip, torch.optim.Adam.state_dict() layout
    ag.experiment_name = "capql_test"
    ag.save(save_dir=str(tmp_path), filename="ck", save_replay_buffer=False)
    ck = th.load(os.path.join(str(tmp_path), "ck.tar"), weights_only=False)
    assert set(ck["policy_state_dict"]) >= {"latent_pi.0.weight", "mean.weight", "log_std_linear.bias", "action_scale"}
    assert "net.4.weight" in ck["q_net_0_state_dict"] and float(ck["q_nets_optimizer_state_dict"]["state"][0]["step"]) == 1
    th.manual_seed(1)
    ag2 = CAPQL(env, net_arch=[32, 32], batch_size=16, buffer_size=256, log=False, seed=1, device=dev, lib=lib)
    ag2.load(os.path.join(str(tmp_path), "ck.tar"), load_replay_buffer=False)
    for name in ("q", "q_target", "pol", "q_exp_avg", "q_exp_avg_sq", "pol_exp_avg"):
        assert th.equal(getattr(ag2.engine, name), getattr(e, name)), name
    assert ag2._q_step == 1 and ag2._p_step == 1


def test_mosac_update_matches_oracle_and_checkpoints(be):
    lib, dev = be
    env = BoxEnv(D=6, Ad=3)
    th.manual_seed(0)
    wts = np.array([0.25, 0.75], dtype=np.float32)
    ag = MOSAC(env, wts, net_arch=[32, 32], batch_size=16, buffer_size=256, log=False, seed=0, device=dev, lib=lib)
    fill_buffer(ag.buffer, 90, 6, 3, 2)
    ag.global_step = 4                                   # % policy_freq == 0 -> actor + alpha updates
    e = ag.engine
    q0 = [cpu(e.q_views(e.q, 0, n)) for n in range(2)]
    tq0 = [cpu(e.q_views(e.q_target, 0, n)) for n in range(2)]
    pol0 = cpu(e.policy_views(e.pol))
    np.random.seed(5)
    th.manual_seed(5)
    snap = rng_snapshot(dev)
    ag.update()
    rng_restore(snap, dev)
    obs, act, rew, nobs, dones = ag.update_inputs()
    eps = th.empty((5, 16, 3), dtype=th.float32, device=dev)       # the agent's draw order: next, (pi_k, alpha_k) per iteration
    eps[0].normal_()
    for k in range(2):
        eps[1 + k].normal_()
        eps[3 + k].normal_()
    eps = eps.cpu()
    qspec, trunk = ac.MlpSpec(9, (32, 32), 2), ac.MlpSpec(6, (32, 32))
    qs = dict(exp_avg=ac.zeros_like(q0[0] + q0[1]), exp_avg_sq=ac.zeros_like(q0[0] + q0[1]))
    ps = dict(exp_avg=ac.zeros_like(pol0), exp_avg_sq=ac.zeros_like(pol0))
    als = dict(exp_avg=[th.zeros(1)], exp_avg_sq=[th.zeros(1)])
    la = th.zeros(1)
    out = ac.mosac_update(qspec, trunk, q0, tq0, pol0, la, qs, ps, als,
                          (obs.cpu(), act.cpu(), rew.cpu(), nobs.cpu(), dones.cpu().reshape(-1, 1)), th.tensor(wts),
                          eps[0], [eps[1], eps[2]], [eps[3], eps[4]], e.action_scale.cpu(), e.action_bias.cpu(),
                          gamma=ag.gamma, tau=ag.tau, q_lr=ag.q_lr, policy_lr=ag.policy_lr, q_step=1, a_step=1,
                          policy_freq=2, do_policy=True, do_target=True, autotune=True, alpha=1.0, target_entropy=-3.0)
    o = ag._out
    assert abs(float(o["q_losses"][0, 0]) - float(out["qf1_loss"])) <= 1e-5 * float(out["qf1_loss"])
    assert abs(float(o["policy_loss"][0]) - float(out["actor_losses"][-1])) <= 1e-5 * abs(float(out["actor_losses"][-1]))
    assert abs(ag.alpha - out["alpha"]) <= 1e-5 * out["alpha"]
    np.testing.assert_allclose(e.log_alpha.cpu().numpy(), la.numpy(), rtol=1e-5, atol=1e-8)
    assert_lists_close(e.policy_views(e.pol_exp_avg), ps["exp_avg"])
    assert_lists_close(e.q_views(e.q_exp_avg, 0, 0) + e.q_views(e.q_exp_avg, 0, 1), qs["exp_avg"])
    assert ag._q_step == 1 and ag._p_step == 2
    # save dict round trip into a fresh learner (MORL/D archives use exactly this)
    sd = ag.get_save_dict(save_replay_buffer=False)
    assert set(sd["actor_state_dict"]) >= {"latent_pi.0.weight", "fc_mean.weight", "fc_logstd.bias"}
    assert "critic.4.bias" in sd["qf1_state_dict"] and "log_alpha" in sd
    th.manual_seed(9)
    ag2 = MOSAC(env, np.array([0.5, 0.5], dtype=np.float32), net_arch=[32, 32], batch_size=16, buffer_size=64,
                log=False, seed=1, device=dev, lib=lib)
    ag2.load(sd, load_replay_buffer=False)
    for name in ("q", "q_target", "pol", "q_exp_avg", "pol_exp_avg_sq", "log_alpha"):
        assert th.equal(getattr(ag2.engine, name), getattr(e, name)), name
    np.testing.assert_array_equal(ag2.weights, wts)
    # odd global step: critics only
    ag.global_step = 5
    pol_before = e.pol.clone()
    ag.update()
    assert th.equal(e.pol, pol_before) and ag._q_step == 2 and ag._p_step == 2


def test_gpils_update_matches_oracle(be):
    lib, dev = be
    env = BoxEnv(D=5, Ad=2, low=-2.0, high=1.0)
    th.manual_seed(0)
    ag = GPILSContinuousAction(env, net_arch=[32, 32], batch_size=8, buffer_size=128, gradient_updates=1, per=True,
                               log=False, seed=0, device=dev, lib=lib, q_drop_rate=0.0)      # LayerNorm on, no dropout
    fill_buffer(ag.replay_buffer, 60, 5, 2, 2, low=-2.0, high=1.0)
    support = [np.array([1.0, 0.0], np.float32), np.array([0.0, 1.0], np.float32), np.array([0.4, 0.6], np.float32)]
    ag.set_weight_support(support)
    e = ag.engine
    q0 = [cpu(e.q_views(e.q, 0, n)) for n in range(2)]
    tq0 = [cpu(e.q_views(e.q_target, 0, n)) for n in range(2)]
    pol0, tpol0 = cpu(e.policy_views(e.pol)), cpu(e.policy_views(e.pol_target))
    weight = th.tensor([0.7, 0.3])
    np.random.seed(2); random.seed(2); th.manual_seed(2)
    snap = rng_snapshot(dev)
    ag.update(weight)
    tree_after = ag.replay_buffer.tree_dev.clone()
    rng_restore(snap, dev)
    # replay the draws: PER indices come from the tree as it was BEFORE the priority update -> rebuild from a copy
    ag2_idx_u = np.random.random_sample(8)
    del ag2_idx_u
    rng_restore(snap, dev)
    qspec = ac.MlpSpec(5 + 2 + 2, (32, 32), 2, layer_norm=True)
    trunk = ac.MlpSpec(5 + 2, (32, 32))
    # the batch the agent used: ask a pristine twin (same buffer contents, untouched tree)
    th.manual_seed(0)
    twin = GPILSContinuousAction(env, net_arch=[32, 32], batch_size=8, buffer_size=128, gradient_updates=1, per=True,
                                 log=False, seed=0, device=dev, lib=lib, q_drop_rate=0.0)
    fill_buffer(twin.replay_buffer, 60, 5, 2, 2, low=-2.0, high=1.0)
    rng_restore(snap, dev)
    s_obs, s_act, s_rew, s_nobs, s_done, idx = twin._sample_batch_experiences()
    ws = [th.tensor(s) for s in support]
    w = th.vstack([weight.expand(8, -1)] + random.choices(ws, k=8))
    noise = th.randn((16, 2), dtype=th.float32, device=dev).cpu()
    batch = [x.cpu().repeat(2, 1) for x in (s_obs, s_act, s_rew, s_nobs, s_done)]
    qs = dict(exp_avg=ac.zeros_like(q0[0] + q0[1]), exp_avg_sq=ac.zeros_like(q0[0] + q0[1]))
    ps = dict(exp_avg=ac.zeros_like(pol0), exp_avg_sq=ac.zeros_like(pol0))
    out = ac.gpipd_cont_update(qspec, trunk, q0, tq0, pol0, tpol0, qs, ps, batch, w, noise, {}, e.action_scale.cpu(),
                               e.action_bias.cpu(), gamma=ag.gamma, lr=ag.learning_rate, tau=ag.tau, q_step=1, p_step=1,
                               do_policy=True, n_per=8)
    assert abs(float(ag._out["critic_loss"][0]) - float(out["critic_loss"])) <= 1e-5 * float(out["critic_loss"])
    assert abs(float(ag._out["policy_loss"][0]) - float(out["policy_loss"])) <= 1e-5 * max(abs(float(out["policy_loss"])), 1e-3)
    assert_lists_close(e.q_views(e.q_exp_avg, 0, 0) + e.q_views(e.q_exp_avg, 0, 1), qs["exp_avg"])
    assert_lists_close(e.policy_views(e.pol_exp_avg), ps["exp_avg"], frac=5e-5)
    assert_lists_close(e.policy_views(e.pol_target), tpol0, rtol=1e-5, frac=1e-6)
    # priorities reached the device tree: leaves of the sampled indices = clip(|td|*0.05 . w, 0.1) ** 0.6
    twin.replay_buffer.update_priorities(idx, th.tensor(out["priority"].astype(np.float32)))
    np.testing.assert_allclose(tree_after.cpu().numpy(), twin.replay_buffer.tree_dev.cpu().numpy(), rtol=2e-5)
    # GPI evaluation: best of the |M| conditioned policies under critic 0
    ag.use_gpi = True
    a = ag.eval(np.linspace(-1, 1, 5).astype(np.float32), np.array([0.5, 0.5], np.float32))
    assert a.shape == (2,) and np.all(a >= -2.0 - 1e-6) and np.all(a <= 1.0 + 1e-6)
    obs_t = th.linspace(-1, 1, 5)[None]
    M = th.stack(ws)
    pol = cpu(e.policy_views(e.pol))
    acts = ac.td3_policy(trunk, pol, obs_t.expand(3, -1), M, e.action_scale.cpu(), e.action_bias.cpu())
    qn = cpu(e.q_views(e.q, 0, 0))
    vals = th.stack([ac.mlp_forward(qspec, qn, th.cat((obs_t.expand(3, -1), acts, M[p].expand(3, -1)), dim=-1))
                     for p in range(3)])                                           # (p, a, R)
    sc = th.einsum("par,r->pa", vals, th.tensor([0.5, 0.5]))
    mq, ai = th.max(sc, dim=1)
    np.testing.assert_allclose(a, acts[ai[th.argmax(mq)]].numpy(), rtol=1e-5, atol=1e-6)


def test_morld_population_update(be):
    lib, dev = be
    env = BoxEnv(D=4, Ad=2)
    th.manual_seed(0)
    np.random.seed(0)
    algo = MORLD(env, pop_size=4, policy_args=dict(net_arch=[16, 16], batch_size=8, buffer_size=64), update_passes=2,
                 shared_buffer=True, exchange_every=10, log=False, seed=3, device=dev, lib=lib,
                 sharing_mechanism=["transfer"])
    assert algo.weights.shape == (4, 2) and np.allclose(algo.weights.sum(1), 1.0)
    # members are slices of the population engine and share one replay buffer
    e = algo.engine
    assert algo.population[2].wrapped.engine.q.data_ptr() == e.q[2].data_ptr()
    assert all(p.wrapped.get_buffer() is algo.population[0].wrapped.get_buffer() for p in algo.population)
    fill_buffer(algo.population[0].wrapped.get_buffer(), 40, 4, 2, 2)
    for p in algo.population:
        p.wrapped.global_step = 10
    before = {k: getattr(e, k).clone() for k in ("q", "pol", "q_target", "log_alpha")}
    cur = algo.population[1]
    algo._update_others(cur)
    if dev.type == "cuda":
        th.cuda.synchronize()
    for i in range(4):
        changed = not th.equal(e.q[i], before["q"][i])
        assert changed == (i != 1), i
        assert (not th.equal(e.pol[i], before["pol"][i])) == (i != 1)
    assert e.q_steps.cpu().tolist() == [2, 0, 2, 2] and e.pol_steps.cpu().tolist() == [4, 0, 4, 4]
    assert [p.wrapped._q_step for p in algo.population] == [2, 0, 2, 2]
    # a member can still advance on its own (its own workspace, its slice of the state, the device step counter)
    cur.wrapped.update()
    assert e.q_steps.cpu().tolist() == [2, 1, 2, 2] and not th.equal(e.q[1], before["q"][1])
    # transfer: the trained actor is copied to the not-yet-trained neighbours
    algo.neighborhoods[1] = [0, 2]
    algo._share(cur)
    assert th.equal(e.pol[2], e.pol[1]) and not th.equal(e.pol[0], e.pol[1]) and int(e.pol_steps[2]) == 0
    # archive pruning through the device Pareto mask
    for k, ev in enumerate([[1.0, 0.0], [0.0, 1.0], [0.4, 0.4], [0.6, 0.6]]):
        algo.archive.add(algo.population[k], np.array(ev))
    got = sorted(tuple(np.round(x, 3)) for x in algo.archive.evaluations)
    assert got == [(0.0, 1.0), (0.6, 0.6), (1.0, 0.0)]


@pytest.mark.parametrize("contexts,pop", [(2, 5), (3, 6)])
def test_morld_population_over_device_contexts_equals_one_population(be, contexts, pop):
    """``MORLD(devices=[...])`` (BASELINE config 5: the population split over the GPUs of a node): G contexts on ONE device give the
    learners of the single population engine bit for bit -- parameters, targets, Adam moments, step counters, entropy coefficients --
    through ``__update_others`` (``morld.py:423-433``) with a member skipped in the middle of a context, and ``_share``'s actor
    transfer across a context boundary; the archive merges the members' evaluations on the host."""
    lib, dev = be
    env = BoxEnv(D=4, Ad=2)

    def make(devices):
        th.manual_seed(0)
        np.random.seed(0)
        random.seed(0)
        if dev.type == "cuda":
            th.cuda.manual_seed(0)
        algo = MORLD(env, pop_size=pop, policy_args=dict(net_arch=[16, 16], batch_size=8, buffer_size=64), update_passes=2,
                     shared_buffer=False, exchange_every=10, log=False, seed=3, device=dev, lib=lib, devices=devices,
                     sharing_mechanism=["transfer"])
        for k, p in enumerate(algo.population):
            fill_buffer(p.wrapped.get_buffer(), 30, 4, 2, 2, seed=k)
            p.wrapped.global_step = 10
        return algo

    one, many = make(None), make([dev] * contexts)
    assert len(one.engines) == 1 and len(many.engines) == contexts
    assert sum(e.pop for e in many.engines) == pop and [g for g, _ in many._where] == sorted(g for g, _ in many._where)
    assert many.population[pop - 1].wrapped.engine.q.data_ptr() == many.engines[-1].q[many._where[pop - 1][1]].data_ptr()

    def state(algo):
        out = {}
        for k in ("q", "q_target", "q_exp_avg", "q_exp_avg_sq", "pol", "pol_exp_avg", "pol_exp_avg_sq", "log_alpha", "q_steps", "pol_steps"):
            out[k] = th.cat([getattr(e, k).reshape(e.pop, -1).cpu() for e in algo.engines])
        return out

    a0, b0 = state(one), state(many)
    assert all(th.equal(a0[k], b0[k]) for k in a0)                       # the members initialise their own slices: same start
    snap = rng_snapshot(dev)                                             # both runs consume the same host / device generators
    one._update_others(one.population[1])
    rng_restore(snap, dev)
    many._update_others(many.population[1])
    if dev.type == "cuda":
        th.cuda.synchronize()
    a1, b1 = state(one), state(many)
    for k in a1:
        assert th.equal(a1[k], b1[k]), k
    assert not th.equal(a1["q"], a0["q"]) and a1["q_steps"].reshape(-1).tolist() == [2, 0] + [2] * (pop - 2)
    # actor transfer across the boundary between two contexts
    src = many._where.index((1, 0)) - 1                                  # last member of context 0 ...
    for algo in (one, many):
        algo.neighborhoods[src] = [src + 1]                              # ... hands its actor to the first member of context 1
        algo._share(algo.population[src])
    a2, b2 = state(one), state(many)
    assert all(th.equal(a2[k], b2[k]) for k in a2) and th.equal(b2["pol"][src + 1], b2["pol"][src])
    for k, ev in enumerate([[1.0, 0.0], [0.0, 1.0], [0.4, 0.4], [0.6, 0.6]]):
        many.archive.add(many.population[k], np.array(ev))
    assert sorted(tuple(np.round(x, 3)) for x in many.archive.evaluations) == [(0.0, 1.0), (0.6, 0.6), (1.0, 0.0)]
    assert many.get_config()["devices"] == [str(dev)] * contexts


def test_simplex_lattice_weights():
    for dim, n in ((2, 6), (3, 6), (3, 64), (4, 10)):
        w = simplex_lattice_weights(dim, n)
        assert w.shape == (n, dim) and np.allclose(w.sum(1), 1.0) and (w >= 0).all()
        assert len({tuple(r) for r in np.round(w, 9)}) == n


def test_short_training_runs(be):
    """train() of every agent steps the environment, fills the device replay and updates without host fallbacks."""
    lib, dev = be
    th.manual_seed(0); np.random.seed(0); random.seed(0)
    env = BoxEnv(D=4, Ad=2)
    env.reward_dim = 2
    ag = CAPQL(env, net_arch=[16, 16], batch_size=8, buffer_size=128, learning_starts=12, log=False, seed=0,
               device=dev, lib=lib)
    p0 = ag.engine.pol.clone()
    ag.train(total_timesteps=20)
    assert ag.global_step == 20 and ag._q_step == 9 and not th.equal(p0, ag.engine.pol)
    ms = MOSAC(BoxEnv(D=4, Ad=2), np.array([0.5, 0.5], np.float32), net_arch=[16, 16], batch_size=8, buffer_size=128,
               learning_starts=10, log=False, seed=0, device=dev, lib=lib)
    ms.train(total_timesteps=18)
    assert ms.global_step == 18 and ms._q_step == 7 and np.isfinite(ms.alpha)
    gp = GPILSContinuousAction(BoxEnv(D=4, Ad=2), net_arch=[16, 16], batch_size=8, buffer_size=128, learning_starts=10,
                               gradient_updates=2, per=True, log=False, seed=0, device=dev, lib=lib)
    sup = [np.array([1.0, 0.0]), np.array([0.0, 1.0])]
    gp.train_iteration(total_timesteps=14, weight=np.array([0.5, 0.5]), weight_support=sup,
                       change_weight_every_episode=True)
    assert gp.global_step == 14 and gp._n_updates == 10 and gp._p_step == 5
    assert np.isfinite(gp.replay_buffer.tree_dev.cpu().numpy()).all()
    # GPI-PD proper: ensemble fit, noisy-policy roll-outs into the model buffer, mixed batches (parity: test_train_traces.py)
    env = BoxEnv(D=4, Ad=2)
    env.spec.id = "mo-halfcheetah-like-box-v0"                                        # an id with a termination rule (never done)
    pd = GPIPDContinuousAction(env, net_arch=[16, 16], batch_size=8, buffer_size=128, learning_starts=10, gradient_updates=2,
                               per=True, log=False, seed=0, device=dev, lib=lib, dynamics_net_arch=[16, 16],
                               dynamics_train_freq=12, dynamics_rollout_len=2, dynamics_rollout_starts=12,
                               dynamics_rollout_freq=6, dynamics_rollout_batch_size=12, dynamics_buffer_size=64,
                               dynamics_min_uncertainty=1e9, dynamics_real_ratio=0.5, dynamics_max_rows=256)
    pd.dynamics_fit_kwargs = dict(max_epochs=3)
    pd.train_iteration(total_timesteps=26, weight=np.array([0.5, 0.5]), weight_support=sup)
    assert pd.global_step == 26 and len(pd.dynamics_buffer) > 0 and pd._last_rollout["imagined"] == 2 * 12
    b = pd._sample_batch_experiences()
    assert b[0].shape[0] == 8 and b[1].shape == (8, 2) and b[5].numel() == 4       # half real (with PER indices), half imagined
    assert np.isfinite(pd.engine.q.cpu().numpy()).all() and np.isfinite(pd._last_holdout)


def test_mosac_discrete_update_and_morld_population(be):
    from morl_baselines_amd.mosac_discrete import MOSACDiscrete
    lib, dev = be
    env = momdp.TreasureLine(0)
    D, A, R = 9, 4, 2
    th.manual_seed(0)
    wts = np.array([0.7, 0.3], dtype=np.float32)
    ag = MOSACDiscrete(env, wts, net_arch=[32, 32], batch_size=16, buffer_size=256, log=False, seed=0, device=dev, lib=lib,
                       update_frequency=2, target_net_freq=4, tau=0.5)
    rng = np.random.default_rng(0)
    for _ in range(80):
        ag.buffer.add(rng.standard_normal(D).astype(np.float32), rng.integers(A), rng.standard_normal(R).astype(np.float32),
                      rng.standard_normal(D).astype(np.float32), rng.random() < 0.1)
    ag.global_step = 8                                   # % target_net_freq == 0 -> Polyak
    e = ag.engine
    q0 = [cpu(e.q_views(e.q, 0, n)) for n in range(2)]
    tq0 = [cpu(e.q_views(e.q_target, 0, n)) for n in range(2)]
    pol0 = cpu(e.policy_views(e.pol))
    np.random.seed(4)
    snap = rng_snapshot(dev)
    ag.update()
    rng_restore(snap, dev)
    obs, act, rew, nobs, dones = ag.update_inputs()
    qspec, pspec = ac.MlpSpec(D, (32, 32), A * R), ac.MlpSpec(D, (32, 32), A)
    qs = dict(exp_avg=ac.zeros_like(q0[0] + q0[1]), exp_avg_sq=ac.zeros_like(q0[0] + q0[1]))
    ps = dict(exp_avg=ac.zeros_like(pol0), exp_avg_sq=ac.zeros_like(pol0))
    als = dict(exp_avg=[th.zeros(1)], exp_avg_sq=[th.zeros(1)])
    la = th.zeros(1)
    out = ac.mosac_discrete_update(qspec, pspec, q0, tq0, pol0, la, qs, ps, als,
                                   (obs.cpu(), act.cpu().reshape(-1, 1), rew.cpu(), nobs.cpu(), dones.cpu().reshape(-1, 1)),
                                   th.tensor(wts), n_actions=A, reward_dim=R, gamma=ag.gamma, tau=0.5, q_lr=ag.q_lr,
                                   policy_lr=ag.policy_lr, step=1, do_target=True, autotune=True, alpha=1.0,
                                   target_entropy=ag.target_entropy)
    o = ag._out
    assert abs(float(o["q_losses"][0, 1]) - float(out["qf2_loss"])) <= 1e-5 * float(out["qf2_loss"])
    assert abs(float(o["policy_loss"][0]) - float(out["actor_loss"])) <= 1e-5 * max(abs(float(out["actor_loss"])), 1e-3)
    assert abs(float(o["alpha_loss"][0]) - float(out["alpha_loss"])) <= 1e-5 * max(abs(float(out["alpha_loss"])), 1e-3)
    np.testing.assert_allclose(e.log_alpha.cpu().numpy(), la.numpy(), rtol=1e-5, atol=1e-8)
    assert_lists_close(e.policy_views(e.pol_exp_avg), ps["exp_avg"], frac=5e-5)
    assert_lists_close(e.q_views(e.q_exp_avg, 0, 0) + e.q_views(e.q_exp_avg, 0, 1), qs["exp_avg"])
    for n in range(2):
        assert_lists_close(e.q_views(e.q_target, 0, n), tq0[n], rtol=1e-5, frac=1e-6)
    a = ag.eval(np.zeros(D, dtype=np.float32))
    assert 0 <= int(a) < A
    sd = ag.get_save_dict()
    assert {"feature_extractor.0.weight", "net.0.weight", "net.2.bias"} <= set(sd["actor_state_dict"])
    # MORL/D with discrete learners: population engine, batched _update_others
    th.manual_seed(1)
    algo = MORLD(momdp.TreasureLine(0), policy_name="MOSACDiscrete", pop_size=3,
                 policy_args=dict(net_arch=[16, 16], batch_size=8, buffer_size=64, update_frequency=2, target_net_freq=4),
                 update_passes=2, shared_buffer=True, exchange_every=10, log=False, seed=3, device=dev, lib=lib)
    buf = algo.population[0].wrapped.get_buffer()
    for _ in range(30):
        buf.add(rng.standard_normal(D).astype(np.float32), rng.integers(A), rng.standard_normal(R).astype(np.float32),
                rng.standard_normal(D).astype(np.float32), rng.random() < 0.1)
    for p in algo.population:
        p.wrapped.global_step = 12
    before = algo.engine.q.clone()
    algo._update_others(algo.population[0])
    assert th.equal(algo.engine.q[0], before[0]) and not th.equal(algo.engine.q[1], before[1])
    assert algo.engine.q_steps.cpu().tolist() == [0, 2, 2] and algo.engine.pol_steps.cpu().tolist() == [0, 2, 2]


def test_gpi_eval_with_64_weight_support(be):
    """BASELINE config 3: the GPI set of 64 weight vectors -- |M|^2 = 4096 critic rows per evaluated action, streamed
    through a batch-128 engine's workspace in chunks; must pick the action the oracle's dense evaluation picks."""
    lib, dev = be
    if dev.type == "cpu":
        pytest.skip("reference-sized networks run on the GPU only (the emulator is slow)")
    D, Ad, R, m = 11, 3, 3, 64                              # mo-hopper-v4 shapes
    env = BoxEnv(D=D, Ad=Ad)
    env.reward_space = momdp.BoxSpace(0.0, 1.0, (R,), 0)
    env.reward_dim = R
    th.manual_seed(3)
    ag = GPILSContinuousAction(env, net_arch=[256, 256], batch_size=128, buffer_size=256, log=False, seed=0, device=dev,
                               lib=lib, q_drop_rate=0.0)
    e = ag.engine
    rng = np.random.default_rng(0)
    sup = [w.astype(np.float32) for w in rng.dirichlet(np.ones(R), m)]
    ag.set_weight_support(sup)
    ag.use_gpi = True
    qspec = ac.MlpSpec(D + Ad + R, (256, 256), R, layer_norm=True, drop_rate=0.0)
    trunk = ac.MlpSpec(D + R, (256, 256))
    pol, qn = cpu(e.policy_views(e.pol)), cpu(e.q_views(e.q, 0, 0))
    M = th.stack([th.tensor(s) for s in sup])
    for trial in range(3):
        obs = rng.standard_normal(D).astype(np.float32)
        w = rng.dirichlet(np.ones(R)).astype(np.float32)
        a = ag.eval(obs, w)
        obs_t = th.tensor(obs)[None]
        acts = ac.td3_policy(trunk, pol, obs_t.expand(m, -1), M, e.action_scale.cpu(), e.action_bias.cpu())
        vals = th.stack([ac.mlp_forward(qspec, qn, th.cat((obs_t.expand(m, -1), acts, M[p].expand(m, -1)), dim=-1))
                         for p in range(m)])
        sc = th.einsum("par,r->pa", vals, th.tensor(w))
        mq, ai = th.max(sc, dim=1)
        best = float(mq.max())
        np.testing.assert_allclose(a, acts[ai[th.argmax(mq)]].numpy(), rtol=1e-4, atol=1e-5)
        # the device's own best value is the oracle's (guards against a near-tie flipping the index silently)
        got_rows = (np.abs(acts.numpy() - a[None]).max(1) < 1e-4).nonzero()[0]
        assert got_rows.size >= 1 and abs(float(sc[:, got_rows[0]].max()) - best) <= 1e-5 * max(1.0, abs(best))


@pytest.mark.parametrize("n_support", [3, 1])
def test_prioritised_loop_in_one_call_equals_sequential_updates(be, n_support):
    """GPI-PD with continuous actions, ``per=True`` (the reference's default): the ``gradient_updates`` loop goes through ONE
    library entry (``morl_ac_update_n_per``: per iteration tree descent + gather, TD3 update, priorities back into the tree) and
    takes exactly the steps of one sample / update / ``update_priorities`` round per iteration
    (gpi_pd_continuous_action.py:373-417).  Parameters and optimiser state to the last bit, the tree to 1e-6 (torch's pow on the
    sequential path, the device's powf inside the tree-update launch on the other)."""
    lib, dev = be
    support = [np.array([1.0, 0.0], np.float32), np.array([0.0, 1.0], np.float32), np.array([0.4, 0.6], np.float32)][:n_support]
    runs = []
    for one_entry in (True, False):
        env = BoxEnv(D=5, Ad=2, low=-2.0, high=1.0)
        th.manual_seed(0)
        ag = GPILSContinuousAction(env, net_arch=[32, 32], batch_size=8, buffer_size=128, gradient_updates=5, per=True,
                                   log=False, seed=0, device=dev, lib=lib)
        ag.per_one_entry_enabled = one_entry
        fill_buffer(ag.replay_buffer, 60, 5, 2, 2, low=-2.0, high=1.0)
        ag.set_weight_support(support)
        np.random.seed(2); random.seed(2); th.manual_seed(2)
        for _ in range(2):
            ag.update(th.tensor([0.7, 0.3]))
        e, b = ag.engine, ag.replay_buffer
        b.flush()
        runs.append((e.q.clone().cpu(), e.pol.clone().cpu(), e.q_target.clone().cpu(), e.q_exp_avg.clone().cpu(),
                     b.tree_dev.clone().cpu(), b.running_max.clone().cpu(), np.random.random_sample(), random.random()))
    a, s = runs
    for k in range(4):
        assert th.equal(a[k], s[k]), k
    for k in (4, 5):
        assert th.allclose(a[k], s[k], rtol=1e-6, atol=0.0), k
    assert a[6] == s[6] and a[7] == s[7]
