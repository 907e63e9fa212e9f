"""Weight-axis sharding of the Envelope step over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device (no distributed code at all); this is the MI355X-native addition of
SURVEY.md 8(e).  Every rank holds a full replica of the agent (parameters, Adam state, replay buffer) and draws the
SAME batch and the SAME W sampled weights (identical seeds => identical host RNG streams).  Rank g owns the weights
``[g*W/G, (g+1)*W/G)``:

  1. it evaluates the next-state slabs Q_online/Q_target(s'_b, w_j) for ITS weights only   (B*W/G rows, both networks
     in one launch pair: ``morl_envelope_slabs``)
  2. ONE all-gather makes both slabs complete on every rank (2 * B*W*A*R*4 bytes in total; 2.4 MB at 256 x 64, 19 MB at
     the weak-scaled 256 x 512 of 8 GPUs); it is issued asynchronously and the training forward of the rank's own rows
     (``morl_envelope_main_forward``, independent of the slabs) runs while it is in flight; the gathered buffer
     [G][2][B][W/G][A][R] is read in place by the TD kernel (no re-layout pass)
  3. envelope arg-max over ALL candidates / TD / backward for its own TD rows (``morl_envelope_update_shard``)
  4. ONE all-reduce sums one flat buffer: gradient | loss | the B PER priorities (non-zero only on the rank that owns
     weight 0, so the sum is exact)                                                              (0.85 MB at [256]*4)
  5. every rank applies the identical clip + Adam step -> replicas stay bit-identical

``shard_envelope_agent(..., axis="batch")`` shards the same step over the BATCH axis instead: rank g keeps the transitions
``[g*B/G, (g+1)*B/G)`` of the common batch and all W weights.  The envelope arg-max of a TD row only looks at the slabs of its
own transition, so nothing has to be gathered: the rank runs the unsharded pipeline (three forward passes in one launch) on its
B/G transitions, normalised by the job's B*W rows, and the ONE collective left is the all-reduce of gradient | loss | priorities.
Same rows per rank as the weight-axis form, one collective and two launches fewer -- this is what ``bench.py`` runs for the
strong-scaled headline; the weight-axis form (the one BASELINE.json's north_star describes) stays for the weak-scaled job, whose
weight axis is the one that grows.

Messages are far below the size where a ring would be bandwidth-bound on the point-to-point xGMI links; they are
latency-bound, so the exchange is kept to two collectives per step and both operate on single contiguous buffers.
The data path has real exchange steps, hence this is *strong* scaling of one 256 x 64 update.

Actor-critic learners (CAPQL, BASELINE config 4) are data-parallel instead: ``shard_capql_agent`` gives every rank its own
rows of the (transition, weight-vector) batch; the critic and the actor gradients are averaged over the ranks inside
``morl_ac_update`` (``morl_ac_cfg.grad_hook`` -> one all-reduce each, 0.3 MB / 0.3 MB at [256, 256]) right before their
Adam steps, so every replica takes the identical step.  A 128-row update is dispatch-latency-bound, so this is *weak*
scaling: per-rank rows stay at ``batch_size`` and the job's batch is ``world * batch_size``.
"""
from __future__ import annotations

import ctypes as C
import os
import types

import numpy as np
import torch as th

from . import ops
from .acnets import randn
from .replay import PrioritizedReplayBuffer
from .envelope import Envelope, random_weights


def _raw_view(ptr: int, count: int, device: th.device) -> th.Tensor:
    """A float32 tensor over ``count`` floats at a raw address the library handed to a transport call-back."""
    if device.type == "cpu":
        return th.from_numpy(np.ctypeslib.as_array((C.c_float * count).from_address(ptr)))

    class _Raw:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return th.as_tensor(_Raw(), device=device)


class NativeComm:
    """The sharded step's collectives behind the C ABI (``morl_comm_*`` / ``morl_allgather_q_begin`` / ``morl_allreduce_grads``
    of include/morl_hip.h).  A communicator is (rank, world) + a TRANSPORT:

    * ``"rccl"``     RCCL over xGMI inside libmorl_hip.so (``morl_comm_init``); ``torch.distributed`` is only the side channel
                     that hands rank 0's unique id to the other ranks
    * ``"torch"``    the caller's transport (``morl_comm_init_custom``): two call-backs that run the all-gather / all-reduce
                     through ``torch.distributed`` on the stream the library hands them -- gloo in the CPU tests (the one-call
                     rank step at world 2 / 4 without a GPU), torch's own RCCL communicator with ``MORL_COMM=torch``
    * ``"loopback"`` one rank, no RCCL (single-rank runs, the emulated build)"""

    def __init__(self, lib, dist, device, group=None, loopback=False, transport="rccl", max_allreduce=0, max_allgather=0):
        self.lib, self.device = lib, th.device(device)
        ident = th.zeros(128, dtype=th.uint8)
        handle = C.c_void_p()
        if loopback:
            # one rank, no RCCL: the all-zero id (morl_comm_init) -- single-rank runs and the emulated build of the CPU tests
            self.rank, self.world, self.transport = 0, 1, "loopback"
            lib.check(lib.lib.morl_comm_init(C.byref(handle), C.c_void_p(ident.data_ptr()), 0, 1))
            self.handle = handle.value
            return
        self.rank, self.world, self.transport = dist.get_rank(group), dist.get_world_size(group), transport
        if transport == "torch":
            self._bind_torch(dist, group, handle)
            return
        if transport == "ipc":
            self._bind_ipc(dist, group, handle, max_allreduce, max_allgather)
            return
        if transport != "rccl":
            raise ValueError("transport must be 'rccl', 'ipc' or 'torch'")
        # Collective-safe id exchange: a rank whose local part fails (rank 0: no RCCL to make an id with; any rank: the communicator
        # refused) still takes part in EVERY collective of the set-up -- the broadcast carries a status byte next to the id, and
        # all ranks agree on the outcome through a MIN all-reduce BEFORE anybody calls ncclCommInitRank (which blocks until every
        # rank has called it) and again after it -- and then all ranks raise together.  make_comm turns that into a common fall-back.
        err = None
        msg = th.zeros(129, dtype=th.uint8)
        if self.rank == 0:
            try:
                lib.check(lib.lib.morl_comm_unique_id(C.c_void_p(ident.data_ptr())))
                msg[:128] = ident
                msg[128] = 1
            except RuntimeError as exc:
                err = exc
        msg = msg.to(self.device)
        dist.broadcast(msg, src=0, group=group)
        msg = msg.cpu()
        have_id = th.tensor([int(msg[128].item())], device=self.device)
        dist.all_reduce(have_id, op=dist.ReduceOp.MIN, group=group)
        if int(have_id.item()) != 1:
            raise RuntimeError(f"RCCL unique id unavailable on rank 0 (this rank: {err})")
        ident = msg[:128].contiguous()
        with (th.cuda.device(self.device) if self.device.type == "cuda" else __import__("contextlib").nullcontext()):
            try:
                lib.check(lib.lib.morl_comm_init(C.byref(handle), C.c_void_p(ident.data_ptr()), self.rank, self.world))
                self.handle = handle.value
            except RuntimeError as exc:
                err = exc
        ok = th.tensor([0 if err is not None else 1], device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) != 1:
            self.close()
            raise RuntimeError(f"RCCL communicator unavailable on at least one rank (this rank: {err})")

    def _bind_ipc(self, dist, group, handle, max_allreduce: int, max_allgather: int) -> None:
        """The single-hop transport (``morl_comm_ipc_*``): this rank's shared region, the world's 64-byte handles exchanged
        through ``torch.distributed`` (the side channel), the peers mapped.  Every step of the set-up is collective-safe: a rank
        whose local part fails still takes part in the exchange and in the final agreement, and then ALL ranks raise together
        (nobody is left waiting in a barrier for a rank that has already given up)."""
        if max_allreduce < 1:
            raise ValueError("the ipc transport needs the size of the largest all-reduce (floats)")
        on_dev = dist.get_backend(group) == "nccl"
        side = self.device if on_dev else th.device("cpu")
        mine = th.zeros(64, dtype=th.uint8)
        err = None
        with (th.cuda.device(self.device) if self.device.type == "cuda" else __import__("contextlib").nullcontext()):
            try:
                self.lib.check(self.lib.lib.morl_comm_ipc_create(C.byref(handle), self.rank, self.world, int(max_allreduce),
                                                                 int(max_allgather), C.c_void_p(mine.data_ptr())))
                self.handle = handle.value
            except RuntimeError as exc:
                err = exc
            every = th.zeros(64 * self.world, dtype=th.uint8, device=side)
            dist.all_gather_into_tensor(every, mine.to(side), group=group)
            every = every.cpu().contiguous()
            if err is None:
                try:
                    self.lib.check(self.lib.lib.morl_comm_ipc_connect(self.handle, C.c_void_p(every.data_ptr())))
                except RuntimeError as exc:
                    err = exc
        ok = th.tensor([0 if err is not None else 1], device=side)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)      # also the barrier: nobody pushes before everybody has mapped
        if int(ok.item()) != 1:
            self.close()
            raise RuntimeError(f"single-hop transport unavailable on at least one rank (this rank: {err})")

    def check(self) -> None:
        """Raises if a bounded wait of the single-hop collectives ran out (a peer never arrived); synchronises the device."""
        self.lib.check(self.lib.lib.morl_comm_check(self.handle))

    def poll(self) -> None:
        """The same verdict over the collectives that have EXECUTED so far, without synchronising (``morl_comm_poll``: a host-mapped
        mirror of the error word).  The sharded training loops call it before every step: a late peer is an exception at most one
        step later, and the step that waited for it was not applied (the clip + Adam launch reads the same word on the device)."""
        self.lib.check(self.lib.lib.morl_comm_poll(self.handle))

    def _bind_torch(self, dist, group, handle) -> None:
        from .native import ALLGATHER_FN, ALLREDUCE_FN
        dev, world, views = self.device, self.world, {}
        self.calls = {"allgather": 0, "allreduce": 0}      # (tests: the one-call path really went through the call-backs)
        self.last_error = None

        def view(ptr, count):
            t = views.get((ptr, count))
            if t is None:                                   # the step's buffers are persistent: a handful of entries
                t = views[(ptr, count)] = _raw_view(ptr, count, dev)
            return t

        def on(stream):
            if dev.type != "cuda":
                import contextlib
                return contextlib.nullcontext()             # the emulated build runs launches synchronously
            return th.cuda.stream(th.cuda.ExternalStream(int(stream or 0), device=dev))

        # gloo with device tensors: both collectives are staged through the host here (D2H on the stream the library hands
        # over, the collective on the host copy, H2D on the same stream).  A test transport -- several ranks sharing ONE GPU run
        # the production rank step at world > 1 on hardware; RCCL refuses duplicate devices.  gloo's own device path for
        # all_reduce (copies on its internal pool streams) was measured to race with the step's kernels when four processes share
        # the device (tools/diag_shared_gpu.py: wrong losses at world 4, right ones with the host staging or with device-wide
        # syncs around the call; profiles/r03_shared_gpu_transport_diag.txt), and it has no device all_gather at all
        host_staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"

        def allgather(_user, send, recv, count, stream):
            try:
                with on(stream):
                    if host_staged:
                        out = th.empty(count * world, dtype=th.float32)
                        dist.all_gather_into_tensor(out, view(send, count).cpu(), group=group)
                        view(recv, count * world).copy_(out)
                    else:
                        dist.all_gather_into_tensor(view(recv, count * world), view(send, count), group=group)
                self.calls["allgather"] += 1
                return 0
            except Exception as exc:                        # never let an exception cross the C frames
                self.last_error = exc
                import traceback
                traceback.print_exc()
                return 1

        def allreduce(_user, buf, count, stream):
            try:
                with on(stream):
                    if host_staged:
                        h = view(buf, count).cpu()
                        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                        view(buf, count).copy_(h)
                    else:
                        dist.all_reduce(view(buf, count), op=dist.ReduceOp.SUM, group=group)
                self.calls["allreduce"] += 1
                return 0
            except Exception as exc:
                self.last_error = exc
                import traceback
                traceback.print_exc()
                return 1

        self._cbs = (ALLGATHER_FN(allgather), ALLREDUCE_FN(allreduce))        # keep the thunks alive with the communicator
        with (th.cuda.device(dev) if dev.type == "cuda" else __import__("contextlib").nullcontext()):
            self.lib.check(self.lib.lib.morl_comm_init_custom(C.byref(handle), self.rank, self.world,
                                                              C.cast(self._cbs[0], C.c_void_p), C.cast(self._cbs[1], C.c_void_p), None))
        self.handle = handle.value

    def size(self):
        """(rank, world) as the LIBRARY's communicator reports them (``morl_comm_size``; over RCCL the world is ``ncclCommCount``)."""
        r, w = C.c_int(-1), C.c_int(-1)
        self.lib.check(self.lib.lib.morl_comm_size(self.handle, C.byref(r), C.byref(w)))
        return r.value, w.value

    def allgather_begin(self, send: th.Tensor, recv: th.Tensor) -> None:
        self.lib.check(self.lib.lib.morl_allgather_q_begin(self.handle, send.data_ptr(), recv.data_ptr(), send.numel(),
                                                           self.lib.stream_of(send)))

    def wait(self, on: th.Tensor) -> None:
        self.lib.check(self.lib.lib.morl_comm_wait(self.handle, self.lib.stream_of(on)))

    def allreduce(self, buf: th.Tensor) -> None:
        self.lib.check(self.lib.lib.morl_allreduce_grads(self.handle, buf.data_ptr(), buf.numel(), self.lib.stream_of(buf)))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.lib.morl_comm_destroy(self.handle)
            self.handle = None
    # (no __del__: tearing a communicator down implicitly at interpreter exit, at different moments on different ranks, is how
    # multi-process jobs hang; the process exit releases it)


def make_comm(lib, dist, device, group=None, transport=None, max_allreduce=0, max_allgather=0):
    """The communicator of a sharded agent.  ``transport``: "rccl" | "ipc" | "torch" | "staged" (None: the ``MORL_COMM`` environment
    variable -- "native" / "rccl", "ipc", "torch", "staged" --, default RCCL on an RCCL process group of the gfx950 build and the
    torch call-backs everywhere else).  If RCCL cannot be brought up inside the library on EVERY rank (librccl missing, a
    second communicator refused) all ranks fall back to the torch transport together and say so on stderr -- the job runs
    instead of dying.  Returns (NativeComm or None for the staged path, name of the transport in use)."""
    import sys
    want = transport or {"native": "rccl"}.get(os.environ.get("MORL_COMM", ""), os.environ.get("MORL_COMM") or None)
    if want is not None and want not in ("rccl", "ipc", "torch", "staged"):
        raise ValueError(f"unknown transport {want!r} (MORL_COMM / transport=): expected 'rccl' ('native'), 'ipc', 'torch' or 'staged'")
    if want == "staged":
        return None, "staged (torch.distributed between library calls)"
    can_rccl = dist.get_backend(group) == "nccl" and lib.is_device_build
    if want is None:
        want = "rccl" if can_rccl else "torch"
    if want == "ipc":
        comm = NativeComm(lib, dist, device, group, transport="ipc", max_allreduce=max_allreduce, max_allgather=max_allgather)
        return comm, "ipc (single-hop direct writes over peer-mapped memory, inside libmorl_hip.so)"
    if want == "rccl":
        if not can_rccl:
            raise ValueError("transport 'rccl' needs an RCCL ('nccl') process group and the gfx950 build")
        # (NativeComm's RCCL set-up is collective-safe: it either succeeds on every rank or raises on every rank, after the
        # ranks have agreed on the outcome -- so the fall-back below is taken by all of them or by none)
        try:
            return NativeComm(lib, dist, device, group, transport="rccl"), "rccl (inside libmorl_hip.so)"
        except RuntimeError as exc:
            err = exc
        print(f"[morl_comm] rank {dist.get_rank(group)}: RCCL inside the library unavailable ({err}); every rank falls back to "
              "the torch.distributed transport", file=sys.stderr, flush=True)
        want = "torch (fallback: morl_comm_init failed)"
    comm = NativeComm(lib, dist, device, group, transport="torch")
    return comm, want + " (torch.distributed call-backs, " + dist.get_backend(group) + ")" if want == "torch" else want


def average_gradients(dist, group=None):
    """``grad_sync`` callable for ``ACEngine.update``: in-place mean over the ranks.  On RCCL the collective runs on its
    own stream; ``wait()`` makes the update's stream wait for it without blocking the host."""
    world = dist.get_world_size(group)
    avg = dist.get_backend(group) == "nccl"            # RCCL averages in the collective; gloo only sums

    def sync(_which: int, grads: th.Tensor):
        work = dist.all_reduce(grads, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True)
        work.wait()
        if not avg:
            grads.mul_(1.0 / world)
    return sync


def shard_capql_agent(agent, dist, group=None):
    """Data-parallel CAPQL: every rank keeps a full replica (identical initial parameters are the caller's job: same seed
    or a broadcast) and ITS OWN replay shard / RNG streams; ``update`` averages the critic and actor gradients over the
    ranks before each Adam step.  The job's batch is ``world * agent.batch_size`` rows."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sync = average_gradients(dist, group)
    agent._shard = types.SimpleNamespace(world=world, rank=rank)
    for buf in (agent.engine.q, agent.engine.pol):         # replicas start from rank 0's parameters
        dist.broadcast(buf, src=0, group=group)
    agent.engine.q_target.copy_(agent.engine.q)

    def update(self):
        e = self.engine
        for _ in range(self.gradient_updates):
            s_obs, s_actions, w, s_rewards, s_next_obs, s_dones = self._sample_batch_experiences()
            B = s_obs.shape[0]
            eps = (randn((B, self.action_dim), e.q.device), randn((B, self.action_dim), e.q.device))
            self._q_step += 1
            self._p_step += 1
            cfg = e.make_cfg(gamma=self.gamma, tau=self.tau, alpha=self.alpha, q_lr=self.learning_rate,
                             policy_lr=self.learning_rate, q_step=self._q_step, policy_step=self._p_step)
            self._out = e.update(cfg, obs=s_obs, actions=s_actions, rewards=s_rewards, next_obs=s_next_obs,
                                 dones=s_dones, w=w, eps_next=eps[0], eps_pi=eps[1], grad_sync=sync)
            self._n_updates += 1

    agent.update = types.MethodType(update, agent)
    return agent


def _rank_step(agent: Envelope, comm, axis: int, offset: int, share: int, slab_loc=None, slab_all=None) -> bool:
    """One sharded iteration of ``agent`` as ONE library entry (``morl_envelope_rank_step``: sampling + the rank step on the agent's
    persistent argument block): the host draws what the reference draws (batch uniforms / indices from the global numpy RNG, the
    sampled weights from ``agent.np_random``) into pinned slots and makes one call.  False: the agent's replay buffer is not one
    of ``replay.py``'s (the caller then takes the call-by-call path)."""
    if agent.per and agent.batch_size > PrioritizedReplayBuffer.TREE_BLOCK:
        # (a prioritised batch larger than one tree-update launch holds: the staged path, whose priority update goes through
        # update_priorities' ascending blocks -- as Envelope.update does for the unsharded agent)
        return False
    st = agent._step_block(1)
    if st is None:
        return False
    W, R = agent.num_sample_w, agent.reward_dim
    ring = agent._w_ring = ops.HostRing.fit(agent.__dict__.get("_w_ring"), agent.lib, agent.device, W * R, th.float32)
    slot, w_ptr = ring.next(W * R)
    slot[:] = random_weights(dim=R, n=W, dist="gaussian", rng=agent.np_random).reshape(-1)
    u_ptr, i_ptr = agent.replay_buffer.draw_batches(agent.batch_size, 1)
    agent._adam_step += 1
    rc = agent.lib.lib.morl_envelope_rank_step(
        agent.q_net.ctx.handle, comm.handle, st.io_ref, axis, offset, share, u_ptr, i_ptr, w_ptr, agent._adam_step,
        float(agent.homotopy_lambda), None if slab_loc is None else slab_loc.data_ptr(), None if slab_all is None else slab_all.data_ptr(),
        agent.lib.stream_of(st.loss))
    ring.mark_used()                 # (also on failure: what was enqueued before it still reads the pinned slots)
    agent.replay_buffer.mark_drawn()
    if rc:
        agent._adam_step -= 1
        agent.lib.check(rc)
    return True


def shard_envelope_agent(agent: Envelope, dist, group=None, emulate=None, comm=None, axis: str = "weights",
                         transport=None) -> Envelope:
    """Replace ``agent.update`` with the sharded step.  ``dist`` is ``torch.distributed`` (already initialised).

    ``emulate=(world, rank)`` is a measurement aid for boxes with one GPU (bench.py --emulate-world): the step of ONE rank of a
    ``world``-rank job with the collectives running among the ranks that really exist -- the other ranks' slabs stay zero and
    nothing is added to the gradient, so the numbers it trains on are meaningless; the kernels, launches, host work and
    message sizes are those of the real job's rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    real_world = world
    if emulate is not None:
        if real_world != 1:
            raise ValueError("emulate= needs a single real rank")
        world, rank = int(emulate[0]), int(emulate[1])
    if axis == "batch":
        return _shard_envelope_batch(agent, dist, group, emulate, comm, world, rank, transport)
    if axis != "weights":
        raise ValueError("axis must be 'weights' or 'batch'")
    W = agent.num_sample_w
    if W % world:
        raise ValueError(f"num_sample_w={W} must be divisible by the number of ranks ({world})")
    Wl = W // world
    i0 = rank * Wl
    A, R = agent.action_dim, agent.reward_dim
    dev = agent.device
    P = agent.q_net.ctx.n_params
    agent._shard = types.SimpleNamespace(world=world, rank=rank, Wl=Wl, i0=i0)
    B0 = agent.batch_size
    # one flat buffer rides in the single all-reduce: [P gradient | 1 loss | B priorities (rank 0's rows of weight 0)]
    agent._grads_x = th.zeros(P + 1 + B0, dtype=th.float32, device=dev)
    agent._grads = agent._grads_x[:P]
    agent._bind_optimizer_state()
    # collectives: inside libmorl_hip.so (RCCL behind the C ABI) on the GPU; torch.distributed itself for the gloo CPU tests
    # (and with MORL_COMM=torch)
    # (``comm``: a ready NativeComm, e.g. the loopback one of a single-rank run)
    if comm is None and hasattr(dist, "broadcast"):          # (a real torch.distributed, not a test stub)
        comm, agent._shard.transport = make_comm(agent.lib, dist, dev, group, transport, max_allreduce=P + 1 + B0,
                                                 max_allgather=2 * B0 * Wl * A * R)
    else:
        agent._shard.transport = "staged" if comm is None else comm.transport
    agent._shard.comm = comm
    slab_loc = th.empty((2, B0, Wl, A, R), dtype=th.float32, device=dev)
    slab_all = th.zeros((world, 2, B0, Wl, A, R), dtype=th.float32, device=dev)
    slab_recv = slab_all if emulate is None else slab_all[rank]

    def update(self: Envelope):
        self._losses = []
        B = self.batch_size
        if B != B0:
            raise ValueError("batch_size changed after shard_envelope_agent")
        if comm is not None:
            comm.poll()                                      # (a timed-out collective of an earlier step: raise, do not train on)
        big_per = self.per and B > PrioritizedReplayBuffer.TREE_BLOCK
        for _ in range(self.gradient_updates):
            gx = self._grads_x
            if comm is not None and _rank_step(self, comm, 1, i0, Wl, slab_loc, slab_all):
                # (sampling + the whole step of this rank: ONE library entry, morl_envelope_rank_step)
                loss = gx[P] if self.gradient_updates == 1 else gx[P].clone()
                pr = gx[P + 1:]
                self._out = {"loss": loss, "priority": pr}
                self._losses.append(loss)
                continue
            aux, sampled_w = self._draw_weights()
            b_obs, b_actions, b_rewards, b_next_obs, b_dones, b_inds = self.replay_buffer.sample(
                B, to_tensor=True, device=self.device, aux=aux,
                prepare=(self.q_net.ctx, self.q_net.flat, self.target_q_net.flat))
            self._w_ring.mark_used()
            ctx = self.q_net.ctx
            self._adam_step += 1
            actions = b_actions.reshape(-1).to(th.int32)
            if comm is not None:
                # the whole step of this rank in one library call (RCCL inside libmorl_hip.so): slabs -> all-gather beside
                # the training forward -> TD / backward -> all-reduce of [gradient | loss | priorities] -> clip + Adam -> PER
                ops.envelope_step_sharded(
                    ctx, comm.handle, self.q_net.flat, self.target_q_net.flat, gx, self._exp_avg, self._exp_avg_sq, b_obs,
                    b_next_obs, actions, b_rewards, b_dones.reshape(-1), sampled_w, i0, Wl, slab_loc, slab_all,
                    gamma=self.gamma, lr=self.learning_rate, adam_step=self._adam_step, max_grad_norm=self.max_grad_norm,
                    homotopy_lambda=float(self.homotopy_lambda), envelope=self.envelope,
                    per=self.replay_buffer.per_update_args(b_inds, self.per_alpha) if (self.per and not big_per) else None)
                if big_per:      # (more entries than the tree update inside the step holds: ascending blocks behind it)
                    self.replay_buffer.update_priorities_from_td(b_inds, gx[P + 1:], self.per_alpha)
            else:
                # the same stages one by one, the collectives through torch.distributed (gloo in the CPU tests)
                w_loc = sampled_w[i0:i0 + Wl].contiguous()
                lazy = ops.shard_lazy(ctx, B, Wl, self.envelope)          # (the form the one-call step takes at this size)
                if lazy:
                    # 1. the ONLINE slab only [B][Wl][A][R]; 2. all-gather -> [G][B][Wl][A][R] (half the message)
                    half = B * Wl * A * R
                    loc = ops.envelope_slab_online(ctx, self.q_net.flat, self.target_q_net.flat, b_next_obs, w_loc,
                                                   slab_loc.view(-1))[:half]
                    recv = slab_all.view(-1)[:world * half] if emulate is None else slab_all.view(-1)[rank * half:(rank + 1) * half]
                    work = dist.all_gather_into_tensor(recv, loc, group=group, async_op=True)
                else:
                    # 1. local slabs [2][B][Wl][A][R]: both networks in one launch pair
                    loc = ops.envelope_slabs(ctx, self.q_net.flat, self.target_q_net.flat, b_next_obs, w_loc, out=slab_loc)
                    # 2. one all-gather -> [G][2][B][Wl][A][R], read in place by the TD kernel; while it is in flight ...
                    work = dist.all_gather_into_tensor(slab_recv.view(-1), loc.view(-1), group=group, async_op=True)
                # ... the training forward of this rank's rows runs (it does not need the slabs)
                ops.envelope_main_forward(ctx, self.q_net.flat, b_obs, w_loc)
                work.wait()
                # 3. this rank's TD rows: arg-max over ALL gathered candidates, TD, backward (priorities: written by the
                #    rank that owns weight 0, zeroed by the others); lazily evaluated: the target network on the selected pairs
                outs = {"loss": gx[P], "priority": gx[P + 1:]}
                flat_all = slab_all.view(-1)
                ops.envelope_update_shard(ctx, self.q_net.flat, self._grads, b_obs, actions, b_rewards, b_dones.reshape(-1),
                                          sampled_w, i0, Wl, flat_all if lazy else slab_all[0, 0], flat_all if lazy else slab_all[0, 1],
                                          gamma=self.gamma, homotopy_lambda=float(self.homotopy_lambda), envelope=self.envelope,
                                          outputs=outs, main_forward_done=True, slab_parts=world,
                                          lazy=(self.target_q_net.flat, b_next_obs) if lazy else None)
                # 4. one all-reduce: flat gradient + loss + priorities
                dist.all_reduce(gx, op=dist.ReduceOp.SUM, group=group)
                # 5. identical optimiser step everywhere
                ops.clip_adam(ctx, self.q_net.flat, self._grads, self._exp_avg, self._exp_avg_sq, lr=self.learning_rate,
                              adam_step=self._adam_step, max_grad_norm=self.max_grad_norm)
                if self.per:
                    self.replay_buffer.update_priorities_from_td(b_inds, gx[P + 1:], self.per_alpha)
            # (views of the all-reduced buffer: valid until the next gradient step overwrites it)
            loss = gx[P] if self.gradient_updates == 1 else gx[P].clone()
            pr = gx[P + 1:]
            self._out = {"loss": loss, "priority": pr}
            self._losses.append(loss)
        # target sync + epsilon / homotopy schedules + logging: the same tail as the single-GPU step
        self._finish_update(pr if self.per else None)

    agent.update = types.MethodType(update, agent)
    agent.q_net.ensure_capacity(agent.batch_size, W)
    return agent


def _shard_envelope_batch(agent: Envelope, dist, group, emulate, comm, world: int, rank: int, transport=None) -> Envelope:
    """Batch-axis sharding (see the module docstring): ``agent.update`` becomes the step of rank ``rank`` of ``world``."""
    B0 = agent.batch_size
    if B0 % world:
        raise ValueError(f"batch_size={B0} must be divisible by the number of ranks ({world})")
    Bl = B0 // world
    b0 = rank * Bl
    dev = agent.device
    P = agent.q_net.ctx.n_params
    agent._shard = types.SimpleNamespace(world=world, rank=rank, Bl=Bl, b0=b0, axis="batch")
    agent._grads_x = th.zeros(P + 1 + B0, dtype=th.float32, device=dev)
    agent._grads = agent._grads_x[:P]
    agent._bind_optimizer_state()
    if comm is None and hasattr(dist, "broadcast"):          # (a real torch.distributed, not a test stub)
        comm, agent._shard.transport = make_comm(agent.lib, dist, dev, group, transport, max_allreduce=P + 1 + B0)
    else:
        agent._shard.transport = "staged" if comm is None else comm.transport
    agent._shard.comm = comm

    def update(self: Envelope):
        self._losses = []
        if self.batch_size != B0:
            raise ValueError("batch_size changed after shard_envelope_agent")
        if comm is not None:
            comm.poll()                                      # (a timed-out collective of an earlier step: raise, do not train on)
        W = self.num_sample_w
        big_per = self.per and B0 > PrioritizedReplayBuffer.TREE_BLOCK
        for _ in range(self.gradient_updates):
            gx = self._grads_x
            if comm is not None and _rank_step(self, comm, 0, b0, Bl):
                # (sampling + the whole step of this rank: ONE library entry, morl_envelope_rank_step)
                loss = gx[P] if self.gradient_updates == 1 else gx[P].clone()
                pr = gx[P + 1:]
                self._out = {"loss": loss, "priority": pr}
                self._losses.append(loss)
                continue
            aux, sampled_w = self._draw_weights()
            b_obs, b_actions, b_rewards, b_next_obs, b_dones, b_inds = self.replay_buffer.sample(
                B0, to_tensor=True, device=self.device, aux=aux,
                prepare=(self.q_net.ctx, self.q_net.flat, self.target_q_net.flat))
            self._w_ring.mark_used()
            ctx = self.q_net.ctx
            self._adam_step += 1
            sl = slice(b0, b0 + Bl)                      # this rank's transitions (row slices of contiguous tensors: views)
            actions = b_actions.reshape(-1).to(th.int32)[sl]
            obs, nobs, rew, done = b_obs[sl], b_next_obs[sl], b_rewards[sl], b_dones.reshape(-1)[sl]
            per = self.replay_buffer.per_update_args(b_inds, self.per_alpha) if (self.per and not big_per) else None
            if comm is not None:
                ops.envelope_step_batch_sharded(
                    ctx, comm.handle, self.q_net.flat, self.target_q_net.flat, gx, self._exp_avg, self._exp_avg_sq, obs, nobs,
                    actions, rew, done, sampled_w, B0, b0, gamma=self.gamma, lr=self.learning_rate, adam_step=self._adam_step,
                    max_grad_norm=self.max_grad_norm, homotopy_lambda=float(self.homotopy_lambda), envelope=self.envelope,
                    per=per)
                if big_per:
                    self.replay_buffer.update_priorities_from_td(b_inds, gx[P + 1:], self.per_alpha)
            else:
                # the same stages through torch.distributed (gloo in the CPU tests): local gradients of the job's loss ...
                gx[P + 1:].zero_()
                outs = {"loss": gx[P], "priority": gx[P + 1 + b0:P + 1 + b0 + Bl], "grad_norm": None}
                ops.envelope_update(ctx, self.q_net.flat, self.target_q_net.flat, self._grads, self._exp_avg, self._exp_avg_sq,
                                    obs, nobs, actions, rew, done, sampled_w, gamma=self.gamma, lr=self.learning_rate,
                                    adam_step=self._adam_step, max_grad_norm=None, homotopy_lambda=float(self.homotopy_lambda),
                                    envelope=self.envelope, apply_step=False, outputs=outs, rows_total=B0 * W)
                # ... one all-reduce, the identical optimiser step everywhere
                dist.all_reduce(gx, op=dist.ReduceOp.SUM, group=group)
                ops.clip_adam(ctx, self.q_net.flat, self._grads, self._exp_avg, self._exp_avg_sq, lr=self.learning_rate,
                              adam_step=self._adam_step, max_grad_norm=self.max_grad_norm)
                if self.per:
                    self.replay_buffer.update_priorities_from_td(b_inds, gx[P + 1:], self.per_alpha)
            loss = gx[P] if self.gradient_updates == 1 else gx[P].clone()
            pr = gx[P + 1:]
            self._out = {"loss": loss, "priority": pr}
            self._losses.append(loss)
        self._finish_update(pr if self.per else None)

    agent.update = types.MethodType(update, agent)
    agent.q_net.ensure_capacity(Bl, agent.num_sample_w)
    return agent
