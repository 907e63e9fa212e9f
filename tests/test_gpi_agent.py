"""GPI-PD / GPI-LS (discrete actions) host mirror: update() against the oracle on the agent's own batch / draws, GPI and
greedy actions, the priority reset over the device records, checkpoints and a short training iteration."""
import os
import random

import numpy as np
import pytest
import torch as th

import gpi_oracle as go
import momdp
from ac_oracle import zeros_like

import morl_baselines_amd.native as native
from morl_baselines_amd.gpi_pd import GPILS, GPIPD


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


def fill(buf, n, D, A, R, seed=0):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        buf.add(rng.standard_normal(D).astype(np.float32), rng.integers(A), rng.standard_normal(R).astype(np.float32),
                rng.standard_normal(D).astype(np.float32), rng.random() < 0.15)


def cpu(vs):
    return [v.detach().cpu().clone() for v in vs]


def test_gpipd_update_actions_reset_priorities_checkpoint(be, tmp_path):
    lib, dev = be
    env = momdp.TreasureLine(0)
    D, A, R = 9, 4, 2
    th.manual_seed(0)
    ag = GPIPD(env, net_arch=[32, 32, 32], batch_size=8, buffer_size=128, learning_starts=10, gradient_updates=3,
               dyna=False, per=True, gpi_pd=True, drop_rate=0.0, log=False, seed=0, device=dev, lib=lib)
    fill(ag.replay_buffer, 50, D, A, R)
    support = [np.array([1.0, 0.0], np.float32), np.array([0.0, 1.0], np.float32), np.array([0.3, 0.7], np.float32)]
    ag.set_weight_support(support)
    e = ag.engine
    spec = go.GpiSpec(D, R, A, (32, 32, 32), True, 0.0)
    q0 = [cpu(e.views(e.q, n)) for n in range(2)]
    tq0 = [cpu(e.views(e.q_target, n)) for n in range(2)]
    # ---- the priority reset over the device records == oracle errors -> tree leaves -----------------------------------------
    wr = th.tensor([0.6, 0.4])
    ag._reset_priorities(wr)
    b = ag.replay_buffer
    err = go.reset_priority_errors(spec, q0, tq0, th.tensor(b.obs[:50]), th.tensor(b.actions[:50].astype(np.float32)),
                                   th.tensor(b.rewards[:50]), th.tensor(b.next_obs[:50]), th.tensor(b.dones[:50]), wr,
                                   [th.tensor(s) for s in support], gamma=ag.gamma, gpi_pd=True)
    leaves = b.tree.nodes[-1][:50]
    np.testing.assert_allclose(leaves, err.clamp(min=0.01).pow(0.6).numpy(), rtol=5e-5)
    # ---- one update() against the oracle on the agent's own draws -------------------------------------------------------------
    ag.global_step = 11                                       # < dynamics_rollout_starts -> a single gradient update
    np.random.seed(3); random.seed(3)
    snap = (random.getstate(), np.random.get_state())
    tree_before = b.tree_dev.clone()
    ag.update(th.tensor([0.5, 0.5]))
    tree_after = b.tree_dev.clone()
    b.tree_dev.copy_(tree_before)                             # re-draw the same indices from the pre-update tree
    random.setstate(snap[0]); np.random.set_state(snap[1])
    s_obs, s_act, s_rew, s_nobs, s_done, idx = ag._sample_batch_experiences()
    weight = th.tensor([0.5, 0.5])
    ws = [th.tensor(s) for s in support]
    w = th.vstack([weight.expand(8, -1)] + random.choices(ws, k=8))
    batch = [x.cpu().repeat(2, 1) for x in (s_obs, s_act.reshape(-1, 1).float(), s_rew, s_nobs, s_done)]
    state = dict(exp_avg=zeros_like(q0[0] + q0[1]), exp_avg_sq=zeros_like(q0[0] + q0[1]))
    out = go.gpi_update(spec, q0, tq0, state, batch, w, th.stack(ws), {}, gamma=ag.gamma, lr=ag.learning_rate, step=1,
                        min_priority=0.01, gpi_pd=True, n_per=8)
    assert abs(ag.last_loss() - float(out["critic_loss"])) <= 1e-5 * float(out["critic_loss"])
    got_m = [v.cpu() for n in range(2) for v in e.views(e.exp_avg, n)]
    for g_, w_ in zip(got_m, state["exp_avg"]):
        np.testing.assert_allclose(g_.numpy(), w_.numpy(), rtol=2e-4, atol=3e-5 * float(w_.abs().max()) + 1e-12)
    b.tree_dev.copy_(tree_before)
    b.update_priorities(idx, th.tensor(out["gpriority"].astype(np.float32)))
    np.testing.assert_allclose(tree_after.cpu().numpy(), b.tree_dev.cpu().numpy(), rtol=3e-5)
    # ---- actions ------------------------------------------------------------------------------------------------------------
    qn = [cpu(e.views(e.q, n)) for n in range(2)]
    rng = np.random.default_rng(1)
    for _ in range(5):
        o = th.tensor(rng.standard_normal(D).astype(np.float32))
        wv = th.tensor([0.2, 0.8])
        assert ag.gpi_action(o, wv, return_policy_index=True) == go.gpi_action(spec, qn[0], o, wv, ws)
        assert ag.max_action(o, wv) == go.max_action(spec, qn, o, wv)
        assert ag.eval(o.numpy(), wv.numpy()) == go.gpi_action(spec, qn[0], o, wv, ws)[0]
    # ---- checkpoint ----------------------------------------------------------------------------------------------------------
    ag.save(save_replay_buffer=False, save_dir=str(tmp_path), filename="g")
    ck = th.load(os.path.join(str(tmp_path), "g.tar"), weights_only=False)
    keys = set(ck["psi_net_0_state_dict"])
    assert {"weights_features.0.weight", "state_features.0.bias", "net.0.weight", "net.1.weight", "net.6.bias"} <= keys
    assert len(ck["M"]) == 3
    th.manual_seed(5)
    ag2 = GPIPD(env, net_arch=[32, 32, 32], batch_size=8, buffer_size=128, dyna=False, drop_rate=0.0, log=False, seed=1,
                device=dev, lib=lib)
    ag2.load(os.path.join(str(tmp_path), "g.tar"), load_replay_buffer=False)
    assert th.equal(ag2.engine.q, e.q) and th.equal(ag2.engine.q_target, e.q) and th.equal(ag2.engine.exp_avg, e.exp_avg)
    assert ag2._adam_step == 1 and len(ag2.weight_support) == 3


def test_gpils_training_iteration(be):
    lib, dev = be
    th.manual_seed(0); np.random.seed(0); random.seed(0)
    env = momdp.TreasureLine(0)
    ag = GPILS(env, net_arch=[16, 16], batch_size=8, buffer_size=256, learning_starts=12, gradient_updates=2, per=True,
               log=False, seed=0, device=dev, lib=lib, target_net_update_freq=5, initial_epsilon=0.5)
    assert ag.gpi_pd is False and ag.dyna is False
    sup = [np.array([1.0, 0.0]), np.array([0.0, 1.0]), np.array([0.5, 0.5])]
    q_before = ag.engine.q.clone()
    ag.train_iteration(total_timesteps=24, weight=np.array([0.5, 0.5]), weight_support=sup)
    assert ag.global_step == 24 and ag._adam_step == 13 and not th.equal(q_before, ag.engine.q)
    assert th.equal(ag.engine.q_target, ag.engine.q) is False or ag.global_step % 5 == 0
    assert np.isfinite(ag.replay_buffer.tree_dev.cpu().numpy()).all()
    ag.train_iteration(total_timesteps=6, weight=np.array([0.2, 0.8]), weight_support=sup, reset_num_timesteps=False)
    assert ag.global_step == 30                                        # second iteration re-prioritised the buffer first


def test_gpipd_dyna_iteration_runs(be):
    """GPI-PD with the Dyna model on either backend: the ensemble is fitted, rollouts fill the model buffer with batched
    adds, updates draw mixed real / imagined batches (step-for-step parity with the reference: test_train_traces.py)."""
    lib, dev = be
    import train_cases as tc
    th.manual_seed(0); np.random.seed(0); random.seed(0)
    env = momdp.TreasureLine(0, env_id=tc.GPIPD_DYNA_ENV_ID)
    ag = GPIPD(env, log=False, seed=0, device=dev, lib=lib, dynamics_train_freq=lambda t: 12, dynamics_max_rows=256,
               **{**tc.GPIPD_DYNA, "dynamics_uncertainty_threshold": 1e9})
    ag.dynamics_fit_kwargs = dict(max_epochs=3)
    ag.train_iteration(total_timesteps=34, weight=tc.WEIGHT.copy(), weight_support=[s.copy() for s in tc.SUPPORT])
    assert ag.global_step == 34 and len(ag.dynamics_buffer) > 0 and ag._last_rollout["imagined"] > 0
    assert np.isfinite(ag._last_holdout) and len(ag.dynamics.elites) == 2
    b = ag._sample_batch_experiences()
    assert b[0].shape[0] == 8 and b[5].numel() == 4                    # real_ratio 0.5: half real (with indices), half imagined
    assert np.isfinite(ag.engine.q.cpu().numpy()).all()


def test_gradient_updates_loop_in_one_call_equals_sequential_updates(be):
    """Without prioritised replay the ``for g in range(self.gradient_updates)`` loop of ``GPIPD.update`` is drawn first and
    submitted as ONE library entry (``morl_gpi_update_n``); it must take exactly the steps of the sequential loop."""
    lib, dev = be
    D, A, R = 9, 4, 2
    support = [np.array([1.0, 0.0], np.float32), np.array([0.0, 1.0], np.float32), np.array([0.3, 0.7], np.float32)]
    runs = []
    for gu, calls in ((3, 1), (1, 3)):
        env = momdp.TreasureLine(0)
        th.manual_seed(0)
        ag = GPILS(env, net_arch=[32, 32, 32], batch_size=8, buffer_size=128, learning_starts=10, gradient_updates=gu,
                   per=False, drop_rate=0.01, log=False, seed=0, device=dev, lib=lib)
        fill(ag.replay_buffer, 50, D, A, R)
        ag.set_weight_support(support)
        ag.global_step = ag.dynamics_rollout_starts + 7          # past the single-update phase; not a target-sync step
        np.random.seed(5); random.seed(5)
        for _ in range(calls):
            ag.update(th.tensor([0.5, 0.5]))
        e = ag.engine
        runs.append((e.q.clone().cpu(), e.exp_avg.clone().cpu(), ag.last_loss(), ag._adam_step))
    a, b = runs
    assert th.equal(a[0], b[0]) and th.equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] == 3


@pytest.mark.parametrize("gpi_pd, n_support", [(True, 3), (False, 3), (True, 1), (True, 7)])
def test_prioritised_loop_in_one_call_equals_sequential_updates(be, gpi_pd, n_support):
    """With prioritised replay (the reference's default) the loop goes through ``morl_gpi_update_n_per``: the host pre-draws the
    uniforms / weight choices in the reference's order, the device samples through the tree, updates and re-prioritises per
    iteration.  Same parameters, optimiser state, tree, running maximum and sampled indices as one ``update`` call per
    iteration (sample -> morl_gpi_update -> update_priorities) (gpi_pd.py:416-420, 507-526).  Parameters and optimiser state to
    the last bit (the same transitions were sampled); the tree to 1e-6: the sequential path raises the priorities to ``alpha``
    with torch's pow, the one-entry path with the device's powf inside the tree-update launch."""
    lib, dev = be
    D, A, R = 9, 4, 2
    rng = np.random.default_rng(3)
    support = [np.array([1.0, 0.0], np.float32), np.array([0.0, 1.0], np.float32)] + \
        [rng.dirichlet(np.ones(2)).astype(np.float32) for _ in range(5)]
    support = support[:n_support]
    runs = []
    for one_entry in (True, False):
        env = momdp.TreasureLine(0)
        th.manual_seed(0)
        ag = GPIPD(env, net_arch=[32, 32, 32], batch_size=8, buffer_size=128, learning_starts=10, gradient_updates=4,
                   dyna=False, per=True, gpi_pd=gpi_pd, drop_rate=0.01, log=False, seed=0, device=dev, lib=lib)
        ag.per_one_entry_enabled = one_entry
        fill(ag.replay_buffer, 50, D, A, R)
        ag.set_weight_support(support)
        ag.global_step = ag.dynamics_rollout_starts + 7          # past the single-update phase; not a target-sync step
        np.random.seed(5); random.seed(5)
        idx_log = []
        for _ in range(2):
            ag.update(th.tensor([0.5, 0.5]))
            if one_entry:
                idx_log.append(ag._last_per_idx.cpu().clone())
        e, b = ag.engine, ag.replay_buffer
        b.flush()
        runs.append((e.q.clone().cpu(), e.exp_avg.clone().cpu(), e.exp_avg_sq.clone().cpu(), b.tree_dev.clone().cpu(),
                     b.running_max.clone().cpu(), ag.last_loss(), ag._adam_step, np.random.random_sample(), random.random(),
                     idx_log))
    a, s = runs
    assert a[6] == s[6] == 8
    for k in range(3):
        assert th.equal(a[k], s[k]), k
    for k in (3, 4):
        assert th.allclose(a[k], s[k], rtol=1e-6, atol=0.0), k
    assert a[5] == s[5] and a[7] == s[7] and a[8] == s[8]       # (both generators left where the sequential loop leaves them)
    assert len(a[9]) == 2 and a[9][0].shape == (4, 8) and int(a[9][0].min()) >= 0 and int(a[9][0].max()) < 50
