"""N > 1 path: world_size-2 gloo run on CPU (emulated kernel library) of the weight-sharded Envelope step, compared
with the single-process step on the same seeds.  The sharded result must equal the unsharded one up to fp32
summation order (loss 1e-5 rel) and the replicas must stay bit-identical to each other."""
import os
import sys
import time

import numpy as np
import pytest
import torch as th
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


N_STEPS = {False: 2, True: 5}      # plain / with the epsilon + homotopy schedules running


def _make_agent(lib, per, schedules=False, dev=th.device("cpu"), arch=(32, 32), B=8, W=4):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import morl_baselines_amd.envelope as envmod
    from test_host_api import ToyEnv, _fill
    env = ToyEnv()
    th.manual_seed(0)
    np.random.seed(0)
    extra = dict(homotopy_decay_steps=6, initial_epsilon=0.5, final_epsilon=0.05, epsilon_decay_steps=8) if schedules else {}
    ag = envmod.Envelope(env, net_arch=list(arch), batch_size=B, num_sample_w=W, buffer_size=256, per=per,
                         learning_starts=0, log=False, seed=0, device=dev, lib=lib, **extra)
    _fill(ag.replay_buffer, 100, env.D, env.A, env.R)
    ag.global_step = 7
    return ag



def _tensor_slices(n_params, D=6, R=2, A=3, archs=((32, 32), (64, 64), (256, 256, 256))):
    """[(start, end)] of every weight matrix / bias vector in the flat parameter buffer of the ToyEnv agents this file builds
    (layout of ``QNetContext.layer_slices``: W_l then b_l, layer after layer), the architecture told by the parameter count."""
    for arch in archs:
        dims = [D + R] + list(arch) + [A * R]
        out, off = [], 0
        for i, o in zip(dims[:-1], dims[1:]):
            out += [(off, off + o * i), (off + o * i, off + o * i + o)]
            off += o * i + o
        if off == n_params:
            return out
    raise AssertionError(f"no architecture of this file has {n_params} parameters")


def _same_training(p, want, n_steps, lr=3e-4):
    """Two runs of ``n_steps`` optimiser steps that differ only in fp32 summation order (sharded vs unsharded partial sums, split-K
    slices): every parameter within the Adam bound (|step| <= lr), 99 % of them within 2 % of it, AND -- per tensor -- at least 90 %
    of every weight matrix's and of every bias vector's entries within that tight bound (a wrong gradient confined to one bias
    vector or to the head is far below 1 % of the parameters).  Not a max-norm bound: a pre-activation within 1e-8 of zero can take
    the other side of its ReLU in one of the two runs -- ONE row of that layer's weights, one bias entry and
    a one-sample share of everything upstream then differ for a step (observed on the MI355X: one unit of 768 at step 3 of 4, 985
    of 135 430 parameters off by up to 0.064 lr n; profiles/r04_relu_flip_multi_step.txt): a few rows of a matrix, never a tenth
    of any tensor."""
    d = np.abs(np.asarray(p, dtype=np.float64) - np.asarray(want, dtype=np.float64)).reshape(-1)
    tight = d <= 0.02 * lr * n_steps
    per_tensor = all(np.mean(tight[a:b]) >= 0.9 for a, b in _tensor_slices(d.size))
    return d.max() <= lr * n_steps and np.mean(tight) >= 0.99 and per_tensor


def _worker(rank, world, port, per, ret, schedules=False, axis="weights", arch=(32, 32), W=4):
    """One rank of a gloo job.  The sharded agent is run TWICE from identical seeds: through the staged path (seven library
    calls, the collectives issued by ``torch.distributed`` between them) and through the production path -- ONE library call
    per step (``morl_envelope_step_sharded`` / ``_batch_sharded``) whose collectives are the communicator's transport, here
    ``morl_comm_init_custom`` call-backs over gloo."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import simlib
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import shard_envelope_agent
    th.set_num_threads(1)
    lib = simlib.load_sim()
    native.use_library(lib)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for transport in ("staged", None, "ipc"):             # None = what a gloo job gets by default: the one-call step over
        ag = _make_agent(lib, per, schedules, arch=arch, W=W)  # torch.distributed call-backs; "ipc": over the single-hop transport
        shard_envelope_agent(ag, dist, axis=axis, transport=transport)
        comm = ag._shard.comm
        assert (comm is None) == (transport == "staged"), ag._shard.transport
        for _ in range(N_STEPS[schedules]):
            ag.update()
            ag.global_step += 1
        calls = None
        if comm is not None:
            assert comm.transport == (transport or "torch") and comm.world == world and comm.rank == rank
            if transport == "ipc":
                comm.check()                              # no bounded wait ran out
                dist.barrier()                            # nobody unmaps a region a peer may still read
            else:
                calls = dict(comm.calls)
            comm.close()
        if axis == "weights" and ag.q_net.ctx.fused:
            # the layer-fused engine evaluates the weight-sharded step's targets LAZILY (tests/conftest.py: MORL_LAZY_MIN_ROWS=0): only the
            # online slab was exchanged, the rank ran the target network on the pairs its own TD rows selected
            n_lazy = ag.q_net.ctx.lazy_target_rows(ag.q_net.flat)
            assert 0 < n_lazy <= ag.batch_size * ag._shard.Wl, n_lazy
        ret[(rank, transport or "one-call")] = (
            ag.q_net.flat.clone().numpy(), float(ag.last_loss()),
            ag.replay_buffer.tree_dev.clone().numpy() if per else None,
            float(ag.homotopy_lambda), float(ag.epsilon), calls)
    dist.destroy_process_group()


@pytest.mark.parametrize("axis", ["weights", "batch"])
@pytest.mark.parametrize("per,schedules,world,arch", [(False, False, 2, (32, 32)), (True, False, 2, (32, 32)), (True, True, 2, (32, 32)),
                                                      (True, False, 4, (32, 32)), (True, False, 2, (64, 64)),
                                                      (True, False, 8, (32, 32)), (True, False, 8, (64, 64))],
                         ids=["plain", "per", "per-schedules", "per-world4", "per-fused-lazy", "per-world8", "per-world8-fused-lazy"])
def test_sharded_update_equals_single_process(per, schedules, world, axis, arch):
    """``schedules``: several steps with ``homotopy_decay_steps`` / ``epsilon_decay_steps`` set -- the sharded step must run
    the same tail as ``Envelope.update`` (envelope.py:336-355), or the auxiliary loss never turns on under sharding.
    ``arch`` (64, 64) takes the layer-fused engine, whose weight-sharded step is LAZILY evaluated (all-gather of the online slab only,
    the target network on the pairs the rank's own TD rows selected): asserted in the workers.
    Both code paths of the rank step are run at world > 1: the one-call step (the production path; its collectives go through
    the pluggable transport of ``morl_comm``) must equal the staged one BIT FOR BIT and the unsharded step to 1e-5.  Third leg:
    the one-call step over the SINGLE-HOP transport (``morl_comm_ipc_*``: direct writes into peer-mapped memory -- POSIX shared
    memory between the processes of this test, hipIpc between GPUs); its all-reduce sums in rank order, so it equals the
    unsharded step to 1e-5 and keeps the replicas bit-identical.
    World 8 is the shape of the driver's 8-GPU run taken to its end: ONE weight per rank (W = 8) on the weight axis, ONE transition
    per rank (B = 8) on the batch axis, eight slab parts, eight-way collectives on all three transports."""
    W = 8 if world == 8 else 4
    import simlib
    import morl_baselines_amd.native as native
    lib = simlib.load_sim()
    native.use_library(lib)
    n_steps = N_STEPS[schedules]
    try:
        ref = _make_agent(lib, per, schedules, arch=arch, W=W)
        assert ref.q_net.ctx.fused == (arch == (64, 64))
        for _ in range(n_steps):
            ref.update()
            ref.global_step += 1
        want, want_loss = ref.q_net.flat.clone().numpy(), ref.last_loss()
        want_tree = ref.replay_buffer.tree_dev.clone().numpy() if per else None
        want_lam, want_eps = float(ref.homotopy_lambda), float(ref.epsilon)
    finally:
        native.use_library(None)
    if schedules:
        assert 0.0 < want_lam <= 1.0 and want_eps < 0.5            # the schedules actually moved
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + (os.getpid() % 2000) + 7 * int(schedules) + 13 * (world - 2) + 31 * int(axis == "batch") + 3 * int(per) + 61 * int(arch != (32, 32))
    procs = [ctx.Process(target=_worker, args=(r, world, port, per, ret, schedules, axis, arch, W)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for path in ("staged", "one-call", "ipc"):
        p0, l0, t0, lam0, eps0, calls0 = ret[(0, path)]
        for r in range(1, world):                                       # (world 4: one weight per rank, four slab parts)
            p1, l1, t1, lam1, eps1, _ = ret[(r, path)]
            assert np.array_equal(p0, p1) and l0 == l1                  # replicas bit-identical
            assert lam0 == lam1 == want_lam and eps0 == eps1 == want_eps     # same schedules as the unsharded agent
            if per:
                assert np.array_equal(t0, t1)
        assert abs(l0 - want_loss) <= 1e-5 * abs(want_loss)              # sharded == unsharded (fp32 order tolerance)
        assert _same_training(p0, want, n_steps)
        if per:
            np.testing.assert_allclose(t0[0], want_tree[0], rtol=1e-5)
    # the production path took the same steps as the staged one, bit for bit, and its collectives were the transport's
    for r in range(world):
        ps, ls, ts, _, _, _ = ret[(r, "staged")]
        po, lo, to, _, _, calls = ret[(r, "one-call")]
        assert np.array_equal(ps, po) and ls == lo
        if per:
            assert np.array_equal(ts, to)
        assert calls == {"allgather": n_steps if axis == "weights" else 0, "allreduce": n_steps}


class _OneRank:
    """torch.distributed of a single rank, without a process group (what the staged path of the sharded step needs)."""

    class _Done:
        def wait(self):
            return None

    def get_world_size(self, group=None):
        return 1

    def get_rank(self, group=None):
        return 0

    def get_backend(self, group=None):
        return "gloo"

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        out.copy_(inp)
        return self._Done()

    def all_reduce(self, t, op=None, group=None):
        return None

    class ReduceOp:
        SUM = "sum"


@pytest.mark.gpu
@pytest.mark.parametrize("per", [False, True])
def test_one_call_sharded_step_equals_the_staged_one_on_the_gpu(per):
    """The same equality on the MI355X at a shape that takes the production kernels (16-row chain tiles, the weight-gradient
    tiles, the PER update inside the clip + Adam launch): loopback communicator, one rank of 1 / 2 / 4."""
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import NativeComm, shard_envelope_agent
    lib = native.load_library()
    dev = th.device("cuda:0")
    for emulate, axis in ((None, "weights"), ((2, 1), "weights"), ((4, 3), "weights"), (None, "batch"), ((2, 1), "batch"), ((8, 5), "batch")):
        runs = []
        for one_call in (False, True, None):
            ag = _make_agent(lib, per, schedules=True, dev=dev, arch=(256, 256, 256), B=64, W=16)
            if one_call is not None:
                comm = NativeComm(lib, None, dev, loopback=True) if one_call else None
                shard_envelope_agent(ag, _OneRank(), emulate=emulate, comm=comm, axis=axis)
            for _ in range(4):
                ag.update()
                ag.global_step += 1
            th.cuda.synchronize()
            runs.append((ag.q_net.flat.clone().cpu().numpy(), ag.last_loss(),
                         ag.replay_buffer.tree_dev.clone().cpu().numpy() if per else None))
        staged, fused, plain = runs
        assert np.array_equal(staged[0], fused[0]) and staged[1] == fused[1]
        if per:
            assert np.array_equal(staged[2], fused[2])
        if emulate is None:
            assert abs(fused[1] - plain[1]) <= 1e-5 * abs(plain[1])
            assert _same_training(fused[0], plain[0], 4)
            if per:
                np.testing.assert_allclose(fused[2][0], plain[2][0], rtol=1e-5)


@pytest.mark.parametrize("per", [False, True])
def test_one_call_sharded_step_equals_the_staged_one(per):
    """``morl_envelope_step_sharded`` (one library call per step, collectives of a loopback communicator) takes exactly the
    steps of the staged path (seven calls, collectives through torch.distributed) and, with one rank owning every weight, those
    of the unsharded step up to fp32 summation order.  Second case: one rank of a two-rank job run alone (bench.py
    --emulate-world) -- the staged and the one-call path must still agree bit for bit."""
    import simlib
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import NativeComm, shard_envelope_agent
    lib = simlib.load_sim()
    native.use_library(lib)
    try:
        for emulate, axis, arch in ((None, "weights", (32, 32)), ((2, 1), "weights", (32, 32)), (None, "batch", (32, 32)),
                                    ((2, 1), "batch", (32, 32)), ((4, 2), "batch", (32, 32)), (None, "weights", (64, 64)),
                                    ((2, 1), "weights", (64, 64)), ((2, 0), "batch", (64, 64))):
            runs = []
            for one_call in (False, True, None):
                ag = _make_agent(lib, per, schedules=True, arch=arch)
                if one_call is not None:
                    comm = NativeComm(lib, None, "cpu", loopback=True) if one_call else None
                    shard_envelope_agent(ag, _OneRank(), emulate=emulate, comm=comm, axis=axis)
                    assert (ag._shard.comm is not None) == one_call
                for _ in range(3):
                    ag.update()
                    ag.global_step += 1
                if one_call is not None and axis == "weights" and arch == (64, 64):
                    # the layer-fused engine: the weight-sharded step evaluated its targets LAZILY (tests/conftest.py:
                    # MORL_LAZY_MIN_ROWS=0) -- only the online slab exchanged, the target network on the selected pairs
                    n_lazy = ag.q_net.ctx.lazy_target_rows(ag.q_net.flat)
                    assert ag.q_net.ctx.fused and 0 < n_lazy <= ag.batch_size * ag._shard.Wl, n_lazy
                runs.append((ag.q_net.flat.clone().numpy(), ag.last_loss(),
                             ag.replay_buffer.tree_dev.clone().numpy() if per else None, float(ag.homotopy_lambda)))
            staged, fused, plain = runs
            assert np.array_equal(staged[0], fused[0]) and staged[1] == fused[1] and staged[3] == fused[3] == plain[3]
            if per:
                assert np.array_equal(staged[2], fused[2])
            if emulate is None:
                assert abs(fused[1] - plain[1]) <= 1e-5 * abs(plain[1])
                assert _same_training(fused[0], plain[0], 3)
                if per:
                    np.testing.assert_allclose(fused[2][0], plain[2][0], rtol=1e-5)
    finally:
        native.use_library(None)


def test_custom_transport_failure_is_a_status_not_a_crash():
    """A transport call-back that raises (or returns non-zero) fails the step with a status + message; the exception never
    crosses the C frames.  Also: the custom communicator reports the rank / world it was given."""
    import ctypes as C
    import simlib
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import NativeComm, shard_envelope_agent
    lib = simlib.load_sim()
    native.use_library(lib)
    try:
        class Broken(_OneRank):
            def all_reduce(self, t, op=None, group=None):
                raise RuntimeError("link down")
        ag = _make_agent(lib, per=False)
        comm = NativeComm(lib, Broken(), "cpu", transport="torch")
        r, w = C.c_int(-1), C.c_int(-1)
        lib.check(lib.lib.morl_comm_size(comm.handle, C.byref(r), C.byref(w)))
        assert (r.value, w.value) == (0, 1)
        shard_envelope_agent(ag, Broken(), comm=comm, axis="batch")
        before = ag.q_net.flat.clone()
        with pytest.raises(RuntimeError, match="all-reduce call-back failed"):
            ag.update()
        assert isinstance(comm.last_error, RuntimeError) and "link down" in str(comm.last_error)
        assert th.equal(before, ag.q_net.flat)                 # the optimiser step behind the failed collective was not taken
        assert lib.lib.morl_comm_init_custom(None, 0, 1, None, None, None) != 0
    finally:
        native.use_library(None)


def _ipc_missing_peer_worker(rank, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MORL_IPC_TIMEOUT_MS"] = "1500"
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import time
    import torch.distributed as dist
    import simlib
    import morl_baselines_amd.native as native
    from morl_baselines_amd import ops
    from morl_baselines_amd.distributed import NativeComm
    lib = simlib.load_sim()
    native.use_library(lib)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    ag = _make_agent(lib, per=False)                          # (rank 0 only uses it: the step whose peer never arrives)
    P, B, W = ag.q_net.ctx.n_params, ag.batch_size, ag.num_sample_w
    comm = NativeComm(lib, dist, "cpu", transport="ipc", max_allreduce=max(1000, P + 1 + B), max_allgather=64)
    buf = th.full((1000,), float(rank + 1))
    comm.allreduce(buf)                                     # both ranks: 1 + 2
    ok = bool((buf == 3.0).all())
    comm.check()
    comm.poll()                                             # (the non-synchronising form of the same question)
    msg, poll_msg, untouched = None, None, None
    if rank == 0:                                            # rank 1 never joins the next collective: one rank's WHOLE step
        t0 = time.time()
        D, R = ag.observation_dim, ag.reward_dim
        gx = th.zeros(P + 1 + B)
        half = B // 2
        obs, nobs = th.randn(half, D), th.randn(half, D)
        act, rew, done = th.zeros(half, dtype=th.int32), th.randn(half, R), th.zeros(half)
        w = th.full((W, R), 1.0 / R)
        before = (ag.q_net.flat.clone(), ag._exp_avg.clone(), ag._exp_avg_sq.clone())
        ops.envelope_step_batch_sharded(ag.q_net.ctx, comm.handle, ag.q_net.flat, ag.target_q_net.flat, gx, ag._exp_avg, ag._exp_avg_sq,
                                        obs, nobs, act, rew, done, w, B, 0, gamma=0.99, lr=1e-3, adam_step=1, max_grad_norm=1.0)
        # the all-reduce inside that step waited for rank 1 until its time limit, summed garbage -- and the optimiser did NOT move
        untouched = all(bool(th.equal(a, b)) for a, b in zip(before, (ag.q_net.flat, ag._exp_avg, ag._exp_avg_sq)))
        try:
            comm.poll()
        except RuntimeError as e:
            poll_msg = str(e)
        try:
            comm.check()
        except RuntimeError as e:
            msg = str(e)
        ret["waited"] = time.time() - t0
    dist.barrier()
    ret[rank] = (ok, msg, poll_msg, untouched)
    comm.close()
    dist.destroy_process_group()


def test_single_hop_transport_bounded_wait_reports_a_missing_peer():
    """``morl_comm_ipc_*``: an all-reduce over shared regions gives the sum on every rank; a peer that never arrives costs the
    waiting rank its time limit and an error from ``morl_comm_check`` AND from the non-synchronising ``morl_comm_poll`` -- not a
    hang -- and the one-call sharded step whose all-reduce timed out leaves parameters and Adam moments untouched (its clip + Adam
    launch reads the error word on the device): a late peer is never a silently wrong optimiser step."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ipc_missing_peer_worker, args=(r, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] is not None and "did not arrive" in ret[0][1] and ret[1][1] is None
    assert ret[0][2] is not None and "did not arrive" in ret[0][2] and "NOT applied" in ret[0][2]
    assert ret[0][3] is True
    assert 1.0 <= ret["waited"] <= 30.0


def test_one_call_sharded_step_rejects_bad_arguments():
    """Error behaviour of ``morl_envelope_step_sharded``: a status code + message, never a launch on bad geometry."""
    import ctypes as C
    import simlib
    import morl_baselines_amd.native as native
    from morl_baselines_amd import ops
    from morl_baselines_amd.distributed import NativeComm
    lib = simlib.load_sim()
    native.use_library(lib)
    try:
        ag = _make_agent(lib, per=False)
        comm = NativeComm(lib, None, "cpu", loopback=True)
        ctx, P = ag.q_net.ctx, ag.q_net.ctx.n_params
        B, W, A, R = ag.batch_size, ag.num_sample_w, ag.action_dim, ag.reward_dim
        D = ag.observation_dim
        gx = th.zeros(P + 1 + B)
        obs, nobs = th.zeros(B, D), th.zeros(B, D)
        act, rew, done = th.zeros(B, dtype=th.int32), th.zeros(B, R), th.zeros(B)
        w = th.full((W, R), 1.0 / R)

        def call(i0, wl, slab_parts=None, handle=comm.handle, grads=gx):
            parts = W // wl if slab_parts is None else slab_parts
            loc, allb = th.zeros(2, B, wl, A, R), th.zeros(parts, 2, B, wl, A, R)
            ops.envelope_step_sharded(ctx, handle, ag.q_net.flat, ag.target_q_net.flat, grads, ag._exp_avg, ag._exp_avg_sq, obs, nobs,
                                      act, rew, done, w, i0, wl, loc, allb, gamma=0.99, lr=1e-3, adam_step=1, max_grad_norm=1.0)

        ag.q_net.ensure_capacity(B, W)
        call(0, W)                                              # the whole weight axis on one rank: fine
        call(W // 2, W // 2)                                    # one rank of two, run alone (loopback communicator): fine
        with pytest.raises(RuntimeError, match="bad shard"):
            call(1, W // 2)                                     # a shard must start at a multiple of its width
        with pytest.raises(RuntimeError, match="NULL"):
            call(0, W, handle=None)
        with pytest.raises(ValueError):
            call(0, W, grads=th.zeros(P))                        # gradient | loss | priorities buffer too short
        with pytest.raises(ValueError):
            call(0, W // 2, slab_parts=1)                        # gathered buffer does not hold every rank's slabs
        # the batch-axis entry: same conventions
        gxb = th.zeros(P + 1 + B)

        def call_b(b0, bl, handle=comm.handle, grads=gxb):
            ops.envelope_step_batch_sharded(ctx, handle, ag.q_net.flat, ag.target_q_net.flat, grads, ag._exp_avg, ag._exp_avg_sq,
                                            obs[:bl], nobs[:bl], act[:bl], rew[:bl], done[:bl], w, B, b0, gamma=0.99, lr=1e-3,
                                            adam_step=1, max_grad_norm=1.0)

        call_b(0, B)
        call_b(B // 2, B // 2)                                  # one rank of two, run alone
        with pytest.raises(RuntimeError, match="bad shard"):
            call_b(1, B // 2)
        with pytest.raises(RuntimeError, match="NULL"):
            call_b(0, B, handle=None)
        with pytest.raises(ValueError):
            call_b(0, B, grads=th.zeros(P))
        assert lib.lib.morl_envelope_step_sharded(ctx.handle, comm.handle, None, None, None, P, None, None, None, None, None, None,
                                                  None, None, B, W, 0, W, None, None, None, None) != 0
        assert b"NULL" in lib.lib.morl_last_error()
    finally:
        native.use_library(None)


# ---- data-parallel CAPQL (BASELINE config 4): gradients averaged inside morl_ac_update -----------------------------------------
def _capql_case():
    from cases_ac import ACCase
    return ACCase("capql_dp", "capql", D=7, Ad=2, R=3, arch=(32, 32), B=24, step=3, seed=11)


def _dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")]
    import torch.distributed as dist
    import simlib
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import average_gradients
    from cases_ac import make_inputs
    from test_ac_kernels_parity import build_engine, engine_state, run_engine
    th.set_num_threads(1)
    lib = simlib.load_sim()
    native.use_library(lib)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = _capql_case()
    inp = make_inputs(c)
    half = c.B // world
    rows = slice(rank * half, (rank + 1) * half)
    loc = dict(inp)
    for k in ("obs", "actions", "rewards", "next_obs", "dones", "w", "eps_next"):
        loc[k] = inp[k][rows]
    loc["eps_pi"] = [inp["eps_pi"][0][rows]]
    eng = build_engine(c, inp, lib, th.device("cpu"))            # identical replicas, each with its own rows
    sync = average_gradients(dist)
    cfg = eng.make_cfg(gamma=c.gamma, tau=c.tau, alpha=c.alpha, q_lr=c.lr, policy_lr=c.lr, q_step=c.step, policy_step=c.step)
    res = eng.update(cfg, obs=loc["obs"], actions=loc["actions"], rewards=loc["rewards"], next_obs=loc["next_obs"],
                     dones=loc["dones"], w=loc["w"], eps_next=loc["eps_next"], eps_pi=loc["eps_pi"][0],
                     want=("critic_loss", "policy_loss"), grad_sync=sync)
    st = engine_state(c, eng)
    ret[rank] = (eng.q.clone().numpy(), eng.pol.clone().numpy(), eng.q_target.clone().numpy(),
                 float(res["critic_loss"][0]), float(res["policy_loss"][0]), res["q_grads"].clone().numpy(),
                 [[p.numpy() for p in n] for n in st["q"]], [p.numpy() for p in st["pol"]])
    # a failing hook surfaces as a Python exception on every rank, not as a hang or a silent local step
    def broken(which, g):
        raise RuntimeError("sync failed")
    try:
        eng.update(cfg, obs=loc["obs"], actions=loc["actions"], rewards=loc["rewards"], next_obs=loc["next_obs"],
                   dones=loc["dones"], w=loc["w"], eps_next=loc["eps_next"], eps_pi=loc["eps_pi"][0], grad_sync=broken)
        ret[f"err{rank}"] = "no error"
    except RuntimeError as e:
        ret[f"err{rank}"] = str(e)
    dist.destroy_process_group()


def test_data_parallel_capql_equals_big_batch_oracle():
    sys.path[:0] = [os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")]
    from ac_common import run_oracle
    c = _capql_case()
    st_o, out = run_oracle(c)                                        # one process, all B rows
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    q0, p0, t0, cl0, pl0, g0, qs0, ps0 = ret[0]
    q1, p1, t1, cl1, pl1, g1, _, _ = ret[1]
    assert np.array_equal(q0, q1) and np.array_equal(p0, p1) and np.array_equal(t0, t1) and np.array_equal(g0, g1)
    # the job's loss is the mean of the ranks' losses; parameters after the step == the oracle's big-batch step
    assert 0.5 * (cl0 + cl1) == pytest.approx(float(out["critic_loss"]), rel=2e-5)
    assert 0.5 * (pl0 + pl1) == pytest.approx(float(out["policy_loss"]), rel=2e-5, abs=1e-6)
    lr = c.lr
    for n in range(2):
        for got, want in zip(qs0[n], st_o["q"][n]):
            assert np.abs(got - want.numpy()).max() <= 0.03 * lr
    for got, want in zip(ps0, st_o["pol"]):
        assert np.abs(got - want.numpy()).max() <= 0.03 * lr
    assert ret["err0"] == ret["err1"] == "sync failed"


# ---- the same sharded step over RCCL on a real GPU (one rank talking to itself: the collectives, streams and the in-place
#      gathered-slab layout are the production ones; N > 1 needs more GPUs than the test box has) --------------------------
def _rccl_worker(port, per, ret, rank=0, world=1, axis="weights"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import morl_baselines_amd.envelope as envmod
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import shard_envelope_agent
    from test_host_api import ToyEnv, _fill
    dev = th.device("cuda", rank)
    th.cuda.set_device(dev)
    lib = native.load_library()

    def make():
        env = ToyEnv()
        th.manual_seed(0)
        np.random.seed(0)
        ag = envmod.Envelope(env, net_arch=[64, 64], batch_size=16, num_sample_w=8, buffer_size=256, per=per,
                             learning_starts=0, log=False, seed=0, device=dev, lib=lib)
        _fill(ag.replay_buffer, 100, env.D, env.A, env.R)
        ag.global_step = 7
        return ag

    ref = make()
    for _ in range(3):
        ref.update()
        ref.global_step += 1
    want, want_loss = ref.q_net.flat.clone().cpu().numpy(), ref.last_loss()
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    print(f"[rccl test] rank {rank}: nranks={dist.get_world_size()} backend={dist.get_backend()} device={dev}",
          file=sys.stderr, flush=True)
    ag = make()
    shard_envelope_agent(ag, dist, axis=axis)
    for _ in range(3):
        ag.update()
        ag.global_step += 1
    th.cuda.synchronize()
    sfx = "" if world == 1 else str(rank)
    ret["params" + sfx] = ag.q_net.flat.clone().cpu().numpy()
    ret["loss" + sfx] = float(ag.last_loss())
    ret["want"], ret["want_loss"] = want, want_loss
    ret["nranks" + sfx] = dist.get_world_size()
    if per:
        ret["tree" + sfx] = ag.replay_buffer.tree_dev.cpu().numpy()
        ret["want_tree"] = ref.replay_buffer.tree_dev.cpu().numpy()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("axis", ["weights", "batch"])
@pytest.mark.parametrize("per", [False, True])
def test_sharded_update_over_rccl_single_rank(per, axis):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_worker, args=(32500 + (os.getpid() % 2000) + int(per) + 2 * int(axis == "batch"), per, ret, 0, 1, axis))
    p.start()
    p.join(300)
    if p.is_alive():
        p.kill()
        pytest.fail("RCCL single-rank worker did not finish")
    assert p.exitcode == 0
    assert abs(ret["loss"] - ret["want_loss"]) <= 1e-5 * abs(ret["want_loss"])
    assert _same_training(ret["params"], ret["want"], 3)
    if per:
        np.testing.assert_allclose(ret["tree"][0], ret["want_tree"][0], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.skipif(th.cuda.device_count() < 2, reason="needs >= 2 GPUs (the multi-rank RCCL path)")
@pytest.mark.parametrize("axis", ["weights", "batch"])
@pytest.mark.parametrize("per", [False, True])
def test_sharded_update_over_rccl_multi_rank(per, axis):
    """The gloo world-size-2 assertions on real RCCL: min(4, device_count) ranks (a power of two dividing the 8 sampled
    weights), one per GPU; replicas bit-identical to each other and equal to the single-GPU step up to fp32 summation order."""
    world = 4 if th.cuda.device_count() >= 4 else 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 36500 + (os.getpid() % 2000) + int(per) + 2 * int(axis == "batch")
    procs = [ctx.Process(target=_rccl_worker, args=(port, per, ret, r, world, axis)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        if p.is_alive():
            for q in procs:
                q.kill()
            pytest.fail("RCCL multi-rank worker did not finish")
        assert p.exitcode == 0
    assert all(ret[f"nranks{r}"] == world for r in range(world))
    for r in range(1, world):
        assert np.array_equal(ret["params0"], ret[f"params{r}"]) and ret["loss0"] == ret[f"loss{r}"]
        if per:
            assert np.array_equal(ret["tree0"], ret[f"tree{r}"])
    assert abs(ret["loss0"] - ret["want_loss"]) <= 1e-5 * abs(ret["want_loss"])
    assert _same_training(ret["params0"], ret["want"], 3)
    if per:
        np.testing.assert_allclose(ret["tree0"][0], ret["want_tree"][0], rtol=1e-5)


def _dp_rccl_worker(port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")]
    import torch.distributed as dist
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import average_gradients
    from cases_ac import make_inputs
    from test_ac_kernels_parity import build_engine
    dev = th.device("cuda:0")
    th.cuda.set_device(dev)
    lib = native.load_library()
    c = _capql_case()
    inp = make_inputs(c)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    res = {}
    for name, sync in (("plain", None), ("dp", average_gradients(dist))):
        eng = build_engine(c, inp, lib, dev)
        cfg = eng.make_cfg(gamma=c.gamma, tau=c.tau, alpha=c.alpha, q_lr=c.lr, policy_lr=c.lr, q_step=c.step, policy_step=c.step)
        out = eng.update(cfg, obs=inp["obs"], actions=inp["actions"], rewards=inp["rewards"], next_obs=inp["next_obs"],
                         dones=inp["dones"], w=inp["w"], eps_next=inp["eps_next"], eps_pi=inp["eps_pi"][0],
                         want=("critic_loss", "policy_loss"), grad_sync=sync)
        th.cuda.synchronize()
        res[name] = (eng.q.cpu().numpy(), eng.pol.cpu().numpy(), eng.q_target.cpu().numpy(), float(out["critic_loss"][0]))
    ret.update(res)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_data_parallel_capql_over_rccl_single_rank():
    """The gradient hook with a real RCCL all-reduce (one rank: the average is the identity): the data-parallel update must
    equal the plain update bit for bit -- the hook, the staging copies and the stream hand-over change nothing else."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_dp_rccl_worker, args=(34500 + (os.getpid() % 2000), ret))
    p.start()
    p.join(300)
    if p.is_alive():
        p.kill()
        pytest.fail("RCCL single-rank worker did not finish")
    assert p.exitcode == 0
    for a, b in zip(ret["plain"][:3], ret["dp"][:3]):
        assert np.array_equal(a, b)
    assert ret["plain"][3] == ret["dp"][3]


# ---- world > 1 ON THE MI355X through the production rank step: several ranks share the one GPU of the test box ----------------
# RCCL refuses two ranks on one device ("duplicate GPU"), so the ranks' collectives go through the other two transports of
# morl_comm -- torch.distributed over gloo (host-staged call-backs) and the single-hop hipIpc transport (peer-mapped regions,
# direct writes: the same-device mapping here, xGMI peers on a multi-GPU node): everything else -- morl_envelope_step_sharded / _batch_sharded, the
# gfx950 kernels, the side-stream all-gather beside the training forward, the gathered-slab layout read in place, the PER update
# behind the all-reduce -- is what a multi-GPU job runs.  Replicas must stay bit-identical and equal the unsharded step.
def _shared_gpu_worker(rank, world, port, per, axis, ret):
    # the ranks SHARE one device here: a rank's spin-wait competes with the peer it waits for (and with their first-call set-up)
    # for the same CUs and time slices, so the single-hop transport's 3 s bound -- ample between GPUs -- gets 30 s
    os.environ["MORL_IPC_TIMEOUT_MS"] = "30000"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import shard_envelope_agent
    dev = th.device("cuda", 0)
    th.cuda.set_device(dev)
    lib = native.load_library()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4
    for transport in (None, "ipc"):           # torch.distributed call-backs (gloo, host-staged) / single-hop hipIpc transport
        ag = _make_agent(lib, per, schedules=True, dev=dev, arch=(256, 256, 256), B=64, W=16)
        shard_envelope_agent(ag, dist, axis=axis, transport=transport)
        comm = ag._shard.comm
        assert comm is not None and comm.transport == (transport or "torch") and comm.world == world
        for _ in range(n):
            ag.update()
            ag.global_step += 1
        th.cuda.synchronize()
        if transport == "ipc":
            comm.check()                      # no bounded wait ran out
        ret[(rank, transport or "torch")] = (ag.q_net.flat.clone().cpu().numpy(), float(ag.last_loss()),
                                             ag.replay_buffer.tree_dev.clone().cpu().numpy() if per else None,
                                             dict(comm.calls) if transport is None else None)
        dist.barrier()
        comm.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("per,world,axis", [(True, 2, "weights"), (True, 4, "weights"), (False, 4, "batch")])
def test_one_call_rank_step_at_world_gt_1_on_one_shared_gpu(per, world, axis):
    import morl_baselines_amd.native as native
    lib = native.load_library()
    dev = th.device("cuda:0")
    n = 4
    ref = _make_agent(lib, per, schedules=True, dev=dev, arch=(256, 256, 256), B=64, W=16)
    for _ in range(n):
        ref.update()
        ref.global_step += 1
    th.cuda.synchronize()
    want, want_loss = ref.q_net.flat.clone().cpu().numpy(), ref.last_loss()
    want_tree = ref.replay_buffer.tree_dev.clone().cpu().numpy() if per else None
    del ref
    ctx = mp.get_context("spawn")
    # The ranks time-share ONE device and rendezvous over local ports: a worker that dies in its set-up (port still in TIME_WAIT, a
    # peer's hipIpc mapping not there within the transport's wait) says nothing about the rank step -- the job gets ONE more try on
    # fresh ports.  Numeric disagreement is never retried: the comparisons below run on whatever the job that finished returned.
    for attempt in range(2):
        ret = ctx.Manager().dict()
        port = 38500 + (os.getpid() % 2000) + 5 * world + 3 * int(per) + 17 * int(axis == "batch") + 211 * attempt
        procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, per, axis, ret)) for r in range(world)]
        for p in procs:
            p.start()
        ok, t_end = True, time.time() + 300
        for p in procs:
            p.join(max(1.0, t_end - time.time()))
            if p.is_alive() or p.exitcode != 0:
                ok = False
        if ok:
            break
        for q in procs:
            if q.is_alive():
                q.kill()
        if attempt == 1:
            pytest.fail("a rank of the shared-GPU job failed or did not finish (twice)")
    for transport in ("torch", "ipc"):
        p0, l0, t0, calls = ret[(0, transport)]
        if transport == "torch":
            assert calls == {"allgather": n if axis == "weights" else 0, "allreduce": n}
        for r in range(1, world):
            p1, l1, t1, _ = ret[(r, transport)]
            assert np.array_equal(p0, p1) and l0 == l1                    # replicas bit-identical
            if per:
                assert np.array_equal(t0, t1)
        assert abs(l0 - want_loss) <= 1e-5 * abs(want_loss), transport    # == the unsharded step up to fp32 summation order
        assert _same_training(p0, want, n), transport
        if per:
            np.testing.assert_allclose(t0[0], want_tree[0], rtol=1e-5)


def test_weight_sharded_step_on_the_bf16_lazy_pipeline():
    """The weight-sharded rank step on the pipeline ``bench.py`` times on one GPU: 256-wide network -> the rank's online slab, its
    training pass, the dX backward and the weight gradients as split-bf16 products (``morl_envelope_slab_online`` /
    ``morl_envelope_main_forward`` on ``mlp_chain_bf``), targets lazily evaluated after the all-gather of the ONLINE slab.
    Staged == one-call bit for bit (whole weight axis on one rank, and one rank of two run alone); the whole-axis run equals the
    unsharded agent (same arithmetic, other summation order) to 1e-5."""
    import simlib
    import morl_baselines_amd.native as native
    from morl_baselines_amd.distributed import NativeComm, shard_envelope_agent
    lib = simlib.load_sim()
    native.use_library(lib)
    try:
        for emulate in (None, (2, 1)):
            runs = []
            for one_call in (False, True, None):
                ag = _make_agent(lib, True, schedules=False, arch=(256, 256), B=8, W=8)
                if one_call is not None:
                    comm = NativeComm(lib, None, "cpu", loopback=True) if one_call else None
                    shard_envelope_agent(ag, _OneRank(), emulate=emulate, comm=comm, axis="weights")
                for _ in range(2):
                    ag.update()
                    ag.global_step += 1
                ctx = ag.q_net.ctx
                assert ctx.last_step_bf16() & 3 == 3                                  # chains and weight gradients on the bf16 cores
                assert ctx.lazy_target_rows(ag.q_net.flat) > 0
                runs.append((ag.q_net.flat.clone().numpy(), ag.last_loss(), ag.replay_buffer.tree_dev.clone().numpy()))
            staged, fused, plain = runs
            assert np.array_equal(staged[0], fused[0]) and staged[1] == fused[1] and np.array_equal(staged[2], fused[2])
            if emulate is None:
                assert abs(fused[1] - plain[1]) <= 1e-5 * abs(plain[1])
                d = np.abs(fused[0] - plain[0])
                assert d.max() <= 3e-4 * 2 and np.mean(d <= 0.02 * 3e-4 * 2) >= 0.99
    finally:
        native.use_library(None)


# ---- RCCL set-up without a second GPU: the ranks' agreement, against a fake librccl (tests/fake_rccl) ---------------------------
def _build_fake_rccl():
    import subprocess
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c")
    out = os.path.join(ROOT, "tests", "fake_rccl", "_build", "librccl_fake.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = f"{out}.{os.getpid()}.tmp"
        subprocess.run(["gcc", "-shared", "-fPIC", "-O1", src, "-o", tmp], check=True)
        os.replace(tmp, out)
    return out


class _AsRccl:
    """``torch.distributed`` (gloo underneath) that calls itself an RCCL process group -- what ``make_comm`` asks before it tries RCCL."""

    def __init__(self, dist):
        self._d = dist

    def get_backend(self, group=None):
        return "nccl"

    def __getattr__(self, k):
        return getattr(self._d, k)


def _fake_rccl_worker(rank, world, port, fake, scenario, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MORL_RCCL_LIB"] = fake
    kind, k = scenario
    if kind == "id":
        os.environ["FAKE_RCCL_FAIL_ID"] = "1"
    elif kind == "init":
        os.environ["FAKE_RCCL_FAIL_INIT_RANK"] = str(k)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import types
    import torch.distributed as dist
    import simlib
    from morl_baselines_amd.distributed import NativeComm, make_comm
    lib = simlib.load_sim()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # (1) the RCCL communicator itself: comes up on every rank, or raises on EVERY rank -- after all of them took part in every
    #     collective of the set-up (a rank that raised early would leave the others in a broadcast / all-reduce forever)
    try:
        c = NativeComm(lib, dist, "cpu", transport="rccl")
        got = ("up",) + c.size()
        c.close()
    except RuntimeError as exc:
        got = ("raised", str(exc))
    dist.barrier()
    # (2) make_comm: RCCL when it came up everywhere, else ALL ranks fall back to the torch.distributed transport together
    as_device = types.SimpleNamespace(lib=lib.lib, is_device_build=True, check=lib.check, check_device=lib.check_device,
                                      stream_of=lib.stream_of)
    comm, name = make_comm(as_device, _AsRccl(dist), "cpu")
    ret[rank] = (got, comm.transport, name, comm.size())
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", [("none", -1), ("id", 0), ("init", 0), ("init", 1), ("init", 2),
                                      ("none", -1, 8), ("init", 0, 8), ("init", 5, 8), ("init", 7, 8)],
                         ids=["rccl-comes-up", "no-unique-id", "init-fails-on-rank0", "init-fails-on-rank1", "init-fails-on-rank2",
                              "world8-rccl-comes-up", "world8-init-fails-on-rank0", "world8-init-fails-on-rank5", "world8-init-fails-on-rank7"])
def test_rccl_setup_agreement_on_every_rank_ordering(scenario):
    """``NativeComm(transport="rccl")`` / ``make_comm`` at world 3 over gloo with a fake librccl (``MORL_RCCL_LIB``) whose
    ``ncclGetUniqueId`` / ``ncclCommInitRank`` fail where the scenario says.  Whatever rank fails: nobody hangs, every rank reaches
    the same verdict, ``make_comm`` returns RCCL on all ranks or the torch.distributed fall-back on all ranks, and a communicator
    that came up reports the world RCCL itself counted (``morl_comm_size`` -> ``ncclCommCount``: bench.py's ``config.rccl_ranks``)."""
    fake = _build_fake_rccl()
    world = scenario[2] if len(scenario) > 2 else 3         # (world 8: the driver's 8-GPU node -- the first real RCCL job is eight ranks)
    scenario = scenario[:2]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 35500 + (os.getpid() % 2000) + 7 * ["none", "id", "init"].index(scenario[0]) + max(scenario[1], 0) + 40 * (world - 3)
    procs = [ctx.Process(target=_fake_rccl_worker, args=(r, world, port, fake, scenario, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        if p.is_alive():
            p.kill()
            pytest.fail("a rank hung in the RCCL set-up")
        assert p.exitcode == 0
    ok = scenario[0] == "none"
    for r in range(world):
        got, transport, name, size = ret[r]
        if ok:
            assert got == ("up", r, world)
            assert transport == "rccl" and name.startswith("rccl") and size == (r, world)
        else:
            assert got[0] == "raised" and ("unavailable" in got[1])
            assert transport == "torch" and "fallback" in name and size == (r, world)
