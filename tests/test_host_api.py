"""Host-side mirror of the reference API on top of the kernels (emulated library on CPU, real one with -m gpu):
replay buffers, the Envelope agent's update() against the oracle, checkpoint round trip, Pareto helpers.
These read like the reference's own tests (tests/test_algos.py::test_envelope, tests/test_pruning.py)."""
import os
import pickle

import numpy as np
import pytest
import torch as th

import envelope_oracle as orc

import morl_baselines_amd.envelope as envmod
import morl_baselines_amd.pareto as par
import morl_baselines_amd.replay as rp
from morl_baselines_amd.native import load_library


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self._n = shape, n
        if n is not None:
            self.n = n
        self._rng = np.random.default_rng(0)

    def sample(self):
        return int(self._rng.integers(self._n)) if self._n is not None else self._rng.standard_normal(self.shape)


class ToyEnv:
    """Tiny MO-Gymnasium-shaped MOMDP (vector obs, discrete actions, vector reward)."""

    def __init__(self, D=6, A=3, R=2, horizon=25, seed=0):
        self.observation_space = _Space(shape=(D,))
        self.action_space = _Space(n=A)
        self.reward_space = _Space(shape=(R,))
        self.unwrapped = self
        self.spec = type("S", (), {"id": "toy-momdp-v0"})()
        self._rng = np.random.default_rng(seed)
        self._M = self._rng.standard_normal((A, D, D)).astype(np.float32) * 0.3
        self._Rw = self._rng.standard_normal((A, D, R)).astype(np.float32)
        self.D, self.A, self.R, self.horizon = D, A, R, horizon

    def reset(self, **kw):
        self.t = 0
        self.s = self._rng.standard_normal(self.D).astype(np.float32)
        return self.s, {}

    def step(self, a):
        self.t += 1
        r = np.tanh(self.s @ self._Rw[a]).astype(np.float32)
        self.s = np.tanh(self.s @ self._M[a] + 0.1 * self._rng.standard_normal(self.D)).astype(np.float32)
        return self.s, r, False, self.t >= self.horizon, {}


@pytest.fixture(scope="module", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    import morl_baselines_amd.native as native
    if request.param == "sim":
        import simlib
        lib = simlib.load_sim()
        native.use_library(lib)          # unpickled buffers re-create themselves through load_library()
        yield lib, th.device("cpu")
        native.use_library(None)
        return
    yield load_library(), th.device("cuda:0")


def _fill(buf, n, D, A, R, seed=0):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        buf.add(rng.standard_normal(D).astype(np.float32), rng.integers(A), rng.standard_normal(R).astype(np.float32),
                rng.standard_normal(D).astype(np.float32), rng.random() < 0.1)


def test_uniform_buffer_matches_reference_semantics(be):
    lib, dev = be
    D, A, R = 5, 4, 3
    buf = rp.ReplayBuffer((D,), 1, rew_dim=R, max_size=64, action_dtype=np.uint8, device=dev, lib=lib)
    _fill(buf, 100, D, A, R)                       # wraps the ring
    assert len(buf) == 64 and buf.ptr == 100 % 64
    np.random.seed(5)
    obs, act, rew, nobs, done, idx = buf.sample(32, to_tensor=True)
    np.random.seed(5)
    want = np.random.choice(64, 32, replace=True)  # buffer.py:82
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(obs.cpu().numpy(), buf.obs[want])
    assert np.array_equal(nobs.cpu().numpy(), buf.next_obs[want])
    assert np.array_equal(rew.cpu().numpy(), buf.rewards[want])
    assert np.array_equal(done.cpu().numpy(), buf.dones[want])
    assert np.array_equal(act.cpu().numpy().astype(np.uint8), buf.actions[want])
    # pickling round trip (Envelope.save stores the buffer object)
    buf2 = pickle.loads(pickle.dumps(buf))
    np.random.seed(6)
    a = buf.sample(16, to_tensor=True)
    np.random.seed(6)
    b = buf2.sample(16, to_tensor=True)
    for x, y in zip(a, b):
        assert th.equal(x.cpu(), y.cpu())


def test_prioritized_buffer_tracks_oracle_tree(be):
    lib, dev = be
    D, A, R, cap = 4, 3, 2, 50
    buf = rp.PrioritizedReplayBuffer((D,), 1, rew_dim=R, max_size=cap, action_dtype=np.uint8, device=dev, lib=lib)
    tree = orc.SumTree(cap)
    minp, ptr = 1e-5, 0
    rng = np.random.default_rng(1)
    for rnd in range(5):
        n = int(rng.integers(5, 30))
        _fill(buf, n, D, A, R, seed=rnd)
        for _ in range(n):
            tree.set(ptr, minp)
            ptr = (ptr + 1) % cap
        np.random.seed(rnd)
        batch = buf.sample(24, to_tensor=True)
        np.random.seed(rnd)
        want = tree.sample(24)
        assert np.array_equal(batch[5].cpu().numpy(), want)
        assert np.array_equal(batch[0].cpu().numpy(), buf.obs[want])
        raw = th.tensor(rng.random(24).astype(np.float32)).to(dev)
        buf.update_priorities_from_td(batch[5], raw, 0.6)
        pr = (raw.cpu().numpy() + np.float32(minp)) ** np.float32(0.6)
        minp = max(minp, pr.max())
        tree.batch_set(want, pr)
        np.testing.assert_allclose(buf.min_priority, float(minp), rtol=3e-7)
        np.testing.assert_allclose(buf.tree.nodes[0][0], tree.nodes[0][0], rtol=1e-6)


def _make_agent(lib, dev, per, **kw):
    env = ToyEnv()
    th.manual_seed(0)
    np.random.seed(0)
    return envmod.Envelope(env, net_arch=[32, 32], batch_size=16, num_sample_w=4, buffer_size=512, per=per,
                           learning_starts=20, log=False, seed=0, device=dev, lib=lib, **kw), env


@pytest.mark.parametrize("per", [False, True])
def test_envelope_update_matches_oracle(be, per):
    """One agent.update() == one oracle step on the very batch / weights the agent drew."""
    lib, dev = be
    ag, env = _make_agent(lib, dev, per)
    _fill(ag.replay_buffer, 200, env.D, env.A, env.R)
    ag.global_step = 21
    params0 = [p.detach().cpu().clone() for p in ag.q_net.ordered_parameters()]
    target0 = [p.detach().cpu().clone() for p in ag.target_q_net.ordered_parameters()]
    # replay the agent's own RNG draws on the host
    st_np = np.random.get_state()
    rng_copy = np.random.default_rng()
    rng_copy.bit_generator.state = ag.np_random.bit_generator.state
    if per:   # the tree is only changed by update(): draw the indices it is about to draw, then rewind the RNG
        idx = ag.replay_buffer.sample_indices(16).cpu().numpy()
    else:
        idx = np.random.choice(ag.replay_buffer.size, 16, replace=True)
    np.random.set_state(st_np)
    ag.update()
    sw = th.tensor(orc.random_weights(env.R, 4, "gaussian", rng=rng_copy)).float()
    b = ag.replay_buffer
    batch = (th.tensor(b.obs[idx]), th.tensor(b.actions[idx]), th.tensor(b.rewards[idx]), th.tensor(b.next_obs[idx]),
             th.tensor(b.dones[idx]))
    m = [th.zeros_like(p) for p in params0]
    v = [th.zeros_like(p) for p in params0]
    o = orc.envelope_update(params0, target0, m, v, 1, batch, sw, n_actions=env.A, reward_dim=env.R, dedup=True)
    assert abs(ag.last_loss() - o["loss"].item()) <= 1e-5 * abs(o["loss"].item())
    for p_dev, p_ref in zip(ag.q_net.ordered_parameters(), params0):
        assert float((p_dev.detach().cpu() - p_ref).abs().max()) <= 0.02 * 3e-4


def test_chain_launch_timing_modes(be):
    """morl_ctx_set_timing: every step (n = 1), every n-th step, one launch per step taking turns (-1), the same on every second
    step (-2), off (0)."""
    lib, dev = be
    env = ToyEnv()
    ag = envmod.Envelope(env, net_arch=[64, 64], batch_size=16, num_sample_w=4, buffer_size=512, per=False, learning_starts=20,
                         log=False, seed=0, device=dev, lib=lib)
    _fill(ag.replay_buffer, 200, env.D, env.A, env.R)
    ag.global_step = 21
    ctx = ag.q_net.ctx
    assert ctx.engine > 0                                   # the layer-fused chain is what is timed
    ctx.set_timing(1)
    ag.update()
    kinds = ctx.read_timing_kinds()
    fwd = kinds["forward"][0] + kinds["forward2"][0]        # ("forward2": the two-pass forward launch of a lazily evaluated step)
    assert fwd >= 1 and kinds["backward"][0] == 1 and kinds["dw"][0] == 1                     # the three GEMM launches
    per_step = sum(n for n, _ in kinds.values())
    chain_per_step = fwd + kinds["backward"][0]
    ctx.set_timing(1)
    ag.update()
    n_chain, ms_chain = ctx.read_timing()                    # the two chain kinds only
    assert n_chain == chain_per_step and ms_chain >= 0.0
    for every, steps, want in ((1, 3, 3 * per_step), (2, 4, 2 * per_step), (-1, 2 * per_step, 2 * per_step),
                               (-2, 4 * per_step, 2 * per_step), (0, 2, 0)):
        ctx.set_timing(every)
        for _ in range(steps):
            ag.update()
        got = ctx.read_timing_kinds()
        assert sum(n for n, _ in got.values()) == want and all(ms >= 0.0 for _, ms in got.values())
        if every < 0:                                        # taking turns: every launch site sampled equally often
            assert got["backward"][0] == 2 and got["dw"][0] == 2 and got["forward"][0] + got["forward2"][0] == 2 * fwd
    assert ctx.read_timing() == (0, 0.0)                    # reading clears the record
    ctx.set_timing(0)
    with pytest.raises(Exception):
        ctx.set_timing(-3)


def test_envelope_train_save_load(be, tmp_path):
    lib, dev = be
    ag, env = _make_agent(lib, dev, per=True)
    ag.train(total_timesteps=40)
    assert ag.global_step == 40 and len(ag.replay_buffer) == 40
    a = ag.eval(np.zeros(env.D, dtype=np.float32), np.array([0.5, 0.5], dtype=np.float32))
    assert 0 <= a < env.A
    ag.save(save_dir=str(tmp_path), filename="ckpt")
    ag2, _ = _make_agent(lib, dev, per=True)
    ag2.load(os.path.join(str(tmp_path), "ckpt.tar"))
    for p, q in zip(ag.q_net.ordered_parameters(), ag2.q_net.ordered_parameters()):
        assert th.equal(p.detach().cpu(), q.detach().cpu())
    for p, q in zip(ag.q_net.ordered_parameters(), ag2.target_q_net.ordered_parameters()):
        assert th.equal(p.detach().cpu(), q.detach().cpu())          # envelope.py:257-258: target := online
    assert th.equal(ag._exp_avg.cpu(), ag2._exp_avg.cpu()) and ag._adam_step == ag2._adam_step
    assert set(th.load(os.path.join(str(tmp_path), "ckpt.tar"), weights_only=False)) == {
        "q_net_state_dict", "q_net_optimizer_state_dict", "replay_buffer"}
    assert list(ag.q_net.state_dict()) == ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias", "net.4.weight",
                                           "net.4.bias"]


def test_envelope_target_signature_matches_reference_method(be):
    """agent.envelope_target(obs, w, sampled_w) with the reference's tiled arguments (envelope.py:284-295)."""
    lib, dev = be
    ag, env = _make_agent(lib, dev, per=False)
    rng = np.random.default_rng(3)
    B, W = 8, 4
    nobs = th.tensor(rng.standard_normal((B, env.D)), dtype=th.float32)
    sw = th.tensor(orc.random_weights(env.R, W, "gaussian", rng=rng), dtype=th.float32)
    w = sw.repeat_interleave(B, 0)
    t_nobs = nobs.repeat(W, 1)
    got = ag.envelope_target(t_nobs.to(dev), w.to(dev), sw.to(dev)).cpu()
    online = [p.detach().cpu() for p in ag.q_net.ordered_parameters()]
    target = [p.detach().cpu() for p in ag.target_q_net.ordered_parameters()]
    want = orc.envelope_target(online, target, t_nobs, w, sw, env.A, env.R)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    got_d = ag.ddqn_target(t_nobs.to(dev), w.to(dev)).cpu()
    want_d, _ = orc.ddqn_target(online, target, t_nobs, w, env.A, env.R)
    assert float((got_d - want_d).abs().max()) <= 1e-5 * float(want_d.abs().max())


# ---- Pareto: the reference's own known-answer generators (tests/test_pruning.py:14-65), restated ---------------------
def _unit_ball_positive(n, dims, rng, lo=0, hi=10):
    pts = np.abs(rng.standard_normal((n, dims)))          # half-normal
    pts = pts / np.sqrt((pts * pts).sum(1))[:, None]
    return pts * (hi - lo) + lo


def _dominated(nd, n, rng):
    dom = rng.choice(nd, size=n, replace=True)
    diffs = rng.uniform(low=0, high=dom)
    active = rng.choice([True, False], size=diffs.shape)
    for row in active:
        if row.sum() == 0:
            row[rng.integers(nd.shape[1])] = True
    return dom - diffs * active


@pytest.mark.parametrize("n_nd,n_d,dims", [(100, 500, 2), (300, 1500, 4)])
def test_filter_pareto_known_front(be, n_nd, n_d, dims):
    lib, dev = be
    if dev.type == "cpu" and n_nd > 100:
        pytest.skip("large front only on the GPU (emulator is O(N^2) fibers)")
    rng = np.random.default_rng(0)
    nd = _unit_ball_positive(n_nd, dims, rng)
    d = _dominated(nd, n_d, rng)
    got = par.filter_pareto_dominated(np.vstack((nd, d)), lib=lib, device=dev)
    assert {tuple(v) for v in got} == {tuple(v) for v in nd}


def test_pareto_archive_matches_reference_semantics(be):
    lib, dev = be
    arch = par.ParetoArchive(lib=lib, device=dev)
    evals = [np.array([1.0, 1.0]), np.array([2.0, 0.5]), np.array([0.5, 0.5]), np.array([2.0, 0.5]),
             np.array([3.0, 3.0])]
    for k, e in enumerate(evals):
        arch.add({"id": k}, e)
    assert [tuple(e) for e in arch.evaluations] == [(3.0, 3.0)] and arch.individuals == [{"id": 4}]
    arch2 = par.ParetoArchive(lib=lib, device=dev)
    for k, e in enumerate(evals[:4]):
        arch2.add({"id": k}, e)
    assert [tuple(e) for e in arch2.evaluations] == [(1.0, 1.0), (2.0, 0.5)]
    assert arch2.individuals == [{"id": 0}, {"id": 1}]
    single = par.filter_pareto_dominated(np.array([[1.0, 2.0]]), lib=lib, device=dev)
    assert single.shape == (1, 2)


@pytest.mark.parametrize("prioritized", [False, True])
def test_add_batch_equals_sequential_adds(be, prioritized):
    """``add_batch`` (the Dyna roll-outs' bulk insert) leaves the buffer -- host arrays, device records, ring pointer and,
    for PER, the sum tree -- exactly as the same transitions added one by one do, including a wrap-around."""
    lib, dev = be
    D, A, R, cap = 4, 3, 2, 40
    cls = rp.PrioritizedReplayBuffer if prioritized else rp.ReplayBuffer
    mk = lambda: cls((D,), 1, rew_dim=R, max_size=cap, action_dtype=np.uint8, device=dev, lib=lib)  # noqa: E731
    one, bulk = mk(), mk()
    rng = np.random.default_rng(4)
    for n in (7, 25, 0, 19, 1):                                   # 52 > cap: the ring wraps inside the 19-block
        o = rng.standard_normal((n, D)).astype(np.float32)
        a = rng.integers(0, A, n)
        r = rng.standard_normal((n, R)).astype(np.float32)
        no = rng.standard_normal((n, D)).astype(np.float32)
        d = rng.random(n) < 0.3
        for k in range(n):
            one.add(o[k], a[k], r[k], no[k], d[k])
        bulk.add_batch(o, a, r, no, d)
        if prioritized and n:                                     # move the running max between blocks
            idx = th.tensor([one.ptr - 1], dtype=th.int64)
            for b in (one, bulk):
                b.update_priorities(idx, np.array([2.5 + n], dtype=np.float32))
    one.flush(); bulk.flush()
    assert (one.ptr, one.size) == (bulk.ptr, bulk.size) == (52 % cap, cap)
    for f in ("obs", "actions", "rewards", "next_obs", "dones"):
        np.testing.assert_array_equal(getattr(one, f), getattr(bulk, f))
    assert th.equal(one.records.cpu(), bulk.records.cpu())
    if prioritized:
        assert th.equal(one.tree_dev.cpu(), bulk.tree_dev.cpu()) and float(one.running_max) == float(bulk.running_max)


def test_priority_update_larger_than_one_launch(be, monkeypatch):
    """``update_priorities`` with more entries than one tree-update launch holds (GPIPD._reset_priorities hands over the
    whole buffer): blocks of the de-duplicated, ascending indices leave the tree bit-identical to the oracle's single
    ``batch_set`` -- duplicates keep their first priority, the running maximum follows."""
    lib, dev = be
    D, A, R, cap = 3, 2, 2, 300
    buf = rp.PrioritizedReplayBuffer((D,), 1, rew_dim=R, max_size=cap, action_dtype=np.uint8, device=dev, lib=lib)
    monkeypatch.setattr(rp.PrioritizedReplayBuffer, "TREE_BLOCK", 64)          # 300 entries -> 5 launches
    rng = np.random.default_rng(2)
    n = 280
    buf.add_batch(rng.standard_normal((n, D)), rng.integers(0, A, n), rng.standard_normal((n, R)),
                  rng.standard_normal((n, D)), rng.random(n) < 0.1)
    tree = orc.SumTree(cap)
    for p in range(n):
        tree.set(p, 1e-5)
    idx = np.concatenate([rng.permutation(n), rng.integers(0, n, 40)])       # every index once, then 40 repeats
    pr = rng.random(idx.size).astype(np.float32) + 0.01
    buf.update_priorities(idx, pr)
    tree.batch_set(idx, pr)
    got = buf.tree
    for lvl in range(len(tree.nodes)):
        np.testing.assert_array_equal(got.nodes[lvl][:len(tree.nodes[lvl])], tree.nodes[lvl])
    assert buf.min_priority == float(max(1e-5, pr.max()))


# ---- round-2 advisor findings, held by tests ---------------------------------------------------------------------------------
def test_pinned_ring_grows_once_and_serves_smaller_batches_from_a_prefix(be):
    """``ops.HostRing``: a buffer that alternates between batch sizes (GPI-PD's real / model mix) keeps ONE ring sized for the
    largest batch -- no re-creation (the pinned block of a dropped ring could be recycled under a kernel still reading it) --
    and the gathers stay right."""
    lib, dev = be
    D, A, R = 5, 4, 3
    buf = rp.ReplayBuffer((D,), 1, rew_dim=R, max_size=64, action_dtype=np.uint8, device=dev, lib=lib)
    _fill(buf, 64, D, A, R)
    rings = []
    for k, B in enumerate((8, 24, 8, 16, 24, 3)):
        np.random.seed(100 + k)
        obs, act, rew, nobs, done, idx = buf.sample(B, to_tensor=True)
        np.random.seed(100 + k)
        want = np.random.choice(64, B, replace=True)
        assert np.array_equal(idx.cpu().numpy(), want) and np.array_equal(obs.cpu().numpy(), buf.obs[want])
        rings.append(buf.__dict__["_idx_ring"])
    assert rings[0] is not rings[1]                                    # 8 -> 24: outgrown once (the old ring retired safely)
    assert all(r is rings[1] for r in rings[1:]) and rings[1].capacity == 24


def test_sharded_steps_refuse_bad_per_arguments_before_the_first_launch(be):
    """A PER batch beyond the tree kernel's 1 024 entries, or missing Adam state, is refused at ENTRY of the one-call rank
    steps: round 2 found them out after the all-reduce, with the parameters already stepped on every rank."""
    lib, dev = be
    from morl_baselines_amd import ops
    from morl_baselines_amd.distributed import NativeComm
    ag, env = _make_agent(lib, dev, per=True)
    _fill(ag.replay_buffer, 100, env.D, env.A, env.R)
    comm = NativeComm(lib, None, dev, loopback=True)
    ctx, P = ag.q_net.ctx, ag.q_net.ctx.n_params
    B, W = 2048, 1
    ag.q_net.ensure_capacity(B, W)
    ctx = ag.q_net.ctx
    gx = th.zeros(P + 1 + B, device=dev)
    z = lambda *s, dt=th.float32: th.zeros(*s, dtype=dt, device=dev)
    w = th.full((W, env.R), 1.0 / env.R, device=dev)
    before = ag.q_net.flat.clone()
    per = ag.replay_buffer.per_update_args(z(B, dt=th.int64), 0.6)
    with pytest.raises(RuntimeError, match="PER update inside the step"):
        ops.envelope_step_batch_sharded(ctx, comm.handle, ag.q_net.flat, ag.target_q_net.flat, gx, ag._exp_avg, ag._exp_avg_sq,
                                        z(B, env.D), z(B, env.D), z(B, dt=th.int32), z(B, env.R), z(B), w, B, 0, gamma=0.99,
                                        lr=1e-3, adam_step=1, max_grad_norm=1.0, per=per)
    with pytest.raises(RuntimeError, match="PER update inside the step"):
        ops.envelope_update(ctx, ag.q_net.flat, ag.target_q_net.flat, gx[:P], ag._exp_avg, ag._exp_avg_sq, z(B, env.D), z(B, env.D),
                            z(B, dt=th.int32), z(B, env.R), z(B), w, gamma=0.99, lr=1e-3, adam_step=1, max_grad_norm=1.0, per=per)
    assert th.equal(before, ag.q_net.flat)                              # nothing ran
    comm.close()


def test_shadow_weights_of_a_skipped_step_are_not_reused(be):
    """``sample(prepare=...)`` makes the K-major shadow copies for the step that follows.  If that step never runs and the
    parameters are then written in place (here: a Polyak copy into the online net), the next forward must not stream the stale
    copies: only the gradient step directly behind ``prepare`` may consume them."""
    lib, dev = be
    from morl_baselines_amd import ops
    ag, env = _make_agent(lib, dev, per=False)
    _fill(ag.replay_buffer, 100, env.D, env.A, env.R)
    ag.replay_buffer.sample(16, to_tensor=True, prepare=(ag.q_net.ctx, ag.q_net.flat, ag.target_q_net.flat))   # step skipped
    new = (ag.q_net.flat * 0.5 + 0.01).clone()
    ops.polyak(lib, new, ag.q_net.flat, 1.0)                             # in-place write behind the library's back
    obs = th.tensor(np.random.default_rng(0).standard_normal((4, env.D)).astype(np.float32)).to(dev)
    w = th.tensor([[0.3, 0.7]], dtype=th.float32).to(dev)
    got = ops.qnet_forward(ag.q_net.ctx, ag.q_net.flat, obs, w).cpu()
    ps, off = [], 0
    for (wo, (o, i), bo, n) in ag.q_net.ctx.layer_slices():
        ps += [new[wo:wo + o * i].view(o, i).cpu(), new[bo:bo + n].cpu()]
    want = orc.qnet_forward(ps, obs.cpu(), w.cpu().expand(4, -1), env.A, env.R)
    assert float((got.view(-1) - want.reshape(-1)).abs().max()) <= 1e-5 * float(want.abs().max())
    ag.q_net.ctx.invalidate_shadows()                                    # (the explicit form is a no-op afterwards)


@pytest.mark.parametrize("per", [False, True])
def test_update_loop_in_one_entry_equals_one_call_per_iteration(be, per):
    """``Envelope.update`` = ``morl_envelope_update_n`` (the whole ``gradient_updates`` loop, sampling included, in ONE library
    call on a persistent argument block) against ``_update_by_calls`` (one ``sample`` + one ``morl_envelope_update`` per iteration):
    same host RNG streams, parameters / Adam state / losses / sum tree / sampled indices equal to the last bit, over several
    calls (the block is reused) and after a change that must rebuild it (``gradient_updates`` grows)."""
    lib, dev = be
    runs = []
    for fast in (True, False):
        ag, env = _make_agent(lib, dev, per, gradient_updates=3)
        _fill(ag.replay_buffer, 200, env.D, env.A, env.R)
        ag.global_step = 21
        losses = []
        for call in range(4):
            if call == 2:
                ag.gradient_updates = 5
            (ag.update if fast else ag._update_by_calls)()
            losses += [float(x) for x in ag._losses]
            ag.global_step += 1
        assert len(losses) == 3 + 3 + 5 + 5 and ag._adam_step == 16
        if fast:
            assert ag._step_state.n_alloc == 5
        runs.append((ag.q_net.flat.clone().cpu(), ag._exp_avg.clone().cpu(), ag._exp_avg_sq.clone().cpu(), losses,
                     ag.replay_buffer.tree_dev.clone().cpu() if per else None, float(ag._out["grad_norm"]),
                     ag._out["priority"].clone().cpu() if per else None, np.random.random_sample(), ag.np_random.random()))
    a, b = runs
    assert th.equal(a[0], b[0]) and th.equal(a[1], b[1]) and th.equal(a[2], b[2])
    assert a[3] == b[3] and a[5] == b[5]
    assert a[7] == b[7] and a[8] == b[8]                     # both host generators were consumed identically
    if per:
        assert th.equal(a[4], b[4]) and th.equal(a[6], b[6])


def test_update_entry_refuses_bad_blocks_before_the_first_launch(be):
    """``morl_envelope_update_n`` validates the whole block up front: a refused call leaves parameters, Adam step and RNG-independent
    state untouched (the agent rewinds its step counter)."""
    lib, dev = be
    ag, env = _make_agent(lib, dev, per=True)
    _fill(ag.replay_buffer, 100, env.D, env.A, env.R)
    ag.global_step = 21
    ag.update()
    before = ag.q_net.flat.clone()
    st = ag._step_state
    keep = st.io.n_levels
    st.io.n_levels = 0                                       # (a corrupted block)
    with pytest.raises(RuntimeError, match="n_levels"):
        ag.update()
    st.io.n_levels = keep
    assert ag._adam_step == 1 and th.equal(ag.q_net.flat, before)
    ag.update()
    assert ag._adam_step == 2 and not th.equal(ag.q_net.flat, before)


def test_prioritised_batch_larger_than_one_tree_update_launch(be):
    """``Envelope.update`` with prioritised replay and a batch of more than 1 024 transitions (one tree-update launch holds 1 024;
    the reference has no such limit, ``envelope.py:329-334``): the step runs without the in-step tree update and the priorities go
    through ``update_priorities``' ascending blocks -- the tree is what ONE ``SumTree.batch_set`` of the reference makes of the same
    priorities, bit for bit, and the running maximum follows ``prioritized_buffer.py:194``."""
    lib, dev = be
    env = ToyEnv()
    th.manual_seed(0)
    np.random.seed(0)
    B = 1100
    ag = envmod.Envelope(env, net_arch=[32, 32], batch_size=B, num_sample_w=2, buffer_size=2048, per=True, learning_starts=0,
                         log=False, seed=0, device=dev, lib=lib)
    _fill(ag.replay_buffer, 1500, env.D, env.A, env.R)
    ag.global_step = 21
    buf = ag.replay_buffer
    buf.flush()
    tree0 = buf.tree_dev.clone().cpu().numpy()
    rmax0 = float(buf.running_max.item())
    st_np = np.random.get_state()
    idx = buf.sample_indices(B).cpu().numpy()                              # what update() is about to draw (the tree is untouched)
    np.random.set_state(st_np)
    ag.update()
    assert ag._adam_step == 1 and np.isfinite(ag.last_loss())
    raw = ag._out["priority"].detach()                       # (the power on the device that took it: a device's powf may differ from
    pr = (raw + th.tensor(rmax0, dtype=th.float32, device=raw.device)).pow(          # the host's by an ulp; the TREE arithmetic is what is exact)
        th.tensor(0.6, dtype=th.float32, device=raw.device)).cpu().numpy()
    want = orc.SumTree(buf.max_size)
    off = 0
    for l, nodes in enumerate(want.nodes):
        nodes[:] = tree0[off:off + len(nodes)]
        off += len(nodes)
    want.batch_set(idx, pr.astype(np.float64))
    got = buf.tree_dev.cpu().numpy()
    off = 0
    for l, nodes in enumerate(want.nodes):
        assert np.array_equal(got[off:off + len(nodes)], nodes), l
        off += len(nodes)
    assert float(buf.running_max.item()) == max(rmax0, float(pr.max()))
    assert len(np.unique(idx)) < B                                          # (duplicates among the sampled leaves: first occurrence wins)
