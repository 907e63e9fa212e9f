"""Envelope-Q TD-update benchmark (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--per 0|1] [--weights 64] [--batch 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` from a bare shell (no WORLD_SIZE in the environment) re-executes itself under
``torch.distributed.run`` with N ranks on 127.0.0.1, one per GPU, and passes rank 0's JSON line through.
N > 1: the headline is STRONG scaling of the metric's fixed 256 x 64 x 3 update, measured on BOTH partitions -- the weight axis
of BASELINE.json's north_star (W/N weights per rank: all-gather of Q(w) + all-reduce) and the batch axis (B/N transitions per
rank: one all-reduce); both figures are in the line (``strong_scaling_axes``), the headline is the faster one and
``config.shard_axis`` names it.  The weak-scaled job (64 weights per GPU, W = 64*N) is measured right after and reported as
the clearly labelled sub-record ``weak_scaling`` (``--scaling weak`` makes it the headline).

At N > 1 the line also carries ``collectives_alone`` (the step's two messages timed by themselves over RCCL and over the single-hop
transport).  ``--gpus N --shared-gpu`` runs the same N > 1 branch with all ranks on cuda:0 (a FUNCTIONAL check of the code path on
a one-GPU box, labelled as such in the line).

One "step" = one ``Envelope.update()`` gradient step (``envelope.py:267-367``) of the HIP agent on the synthetic
workload of BASELINE.md section 3: obs dim 32, 3 objectives, 6 actions, net [256]*4, batch 256 x 64 sampled weights
(16 384 TD rows = 49 152 scalar TD errors per step), replay buffer pre-filled with 20 000 seeded transitions, PER on
(the reference default).  The replay data is resident in HBM before the timed region starts; the per-step host work
(index / weight sampling on the reference's RNG streams) is inside it.
Prints ONE JSON line (rank 0).  ``value`` = TD-row updates per second of the whole job.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch as th

D, A, R = 32, 6, 3
ARCH = [256, 256, 256, 256]
MACS_ROW = (D + R) * 256 + 3 * 256 * 256 + 256 * A * R           # 210 176 MACs per row-forward
FWD_FLOP_ROW = 2 * MACS_ROW                                       # 420 352
BWD_DX_FLOP_ROW = 2 * (A * R * 256 + 3 * 256 * 256)               # dX chain (no dX for layer 0): 402 432
PEAK_FP32_MFMA_TFLOPS = 157.3                                     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0                                    # MI355X_MICROARCH.md: dense bf16 MFMA (no 2:1 sparsity)
BF16_PRODUCTS = 6                                                 # split-bf16 products per fp32 product (csrc/mlp_chain_bf.h)
ALGORITHMIC_BYTES_STEP = 7.7e6                                    # batch + parameters + Adam state + outputs of one step
def emulated_ceiling():
    """One rank of an N-rank strong-scaled job run alone on one MI355X (--force-shard --emulate-world N): single-GPU step time / that
    rank's step time = what N GPUs could reach if the collectives were free.  Read from the newest committed summary
    (profiles/r*_emulated_ceiling.json, written by tools/emulated_ceiling.py from the bench lines of the same build) -- a constant of
    the repository, not of this run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_emulated_ceiling.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    d["source"] = os.path.relpath(files[-1], ROOT)
    d.setdefault("note", "upper bounds BEFORE any collective latency; the >= 6x of north_star is only reachable in the weak reading "
                         "(W grows with N), which is a different workload from the metric")
    return d


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape = shape
        if n is not None:
            self.n = n
        self._rng = np.random.default_rng(0)

    def sample(self):
        return int(self._rng.integers(self.n))


class SyntheticEnv:
    """Spaces only (the benchmark never steps an environment)."""

    def __init__(self):
        self.observation_space = _Space(shape=(D,))
        self.action_space = _Space(n=A)
        self.reward_space = _Space(shape=(R,))
        self.unwrapped = self
        self.spec = type("S", (), {"id": "synthetic-minecart-like-v0"})()


def fill_buffer(buf, n, seed=0):
    """BASELINE.md section 3 draw order: obs, action, reward, next_obs, done per transition."""
    rng = np.random.default_rng(seed)
    for _ in range(n):
        obs = rng.standard_normal(D).astype(np.float32)
        action = rng.integers(A)
        reward = rng.standard_normal(R).astype(np.float32)
        next_obs = rng.standard_normal(D).astype(np.float32)
        done = rng.random() < 0.05
        buf.add(obs, action, reward, next_obs, done)


def cpu_baseline(batch, weights, per, budget_s=25.0):
    """The oracle (a line-by-line port of the reference's as-written W^2*B-row update) timed on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import envelope_oracle as orc

    th.manual_seed(0)
    rng = np.random.default_rng(0)
    online = orc.init_qnet_params(D, A, R, ARCH)
    target = [p.clone() for p in online]
    m = [th.zeros_like(p) for p in online]
    v = [th.zeros_like(p) for p in online]
    mk = lambda: (th.tensor(rng.standard_normal((batch, D)), dtype=th.float32),
                  th.tensor(rng.integers(A, size=(batch, 1)), dtype=th.uint8),
                  th.tensor(rng.standard_normal((batch, R)), dtype=th.float32),
                  th.tensor(rng.standard_normal((batch, D)), dtype=th.float32),
                  th.tensor((rng.random((batch, 1)) < 0.05), dtype=th.float32))
    step = [0]

    def one():
        sw = th.tensor(orc.random_weights(R, weights, "gaussian", rng=rng), dtype=th.float32)
        step[0] += 1
        t0 = time.perf_counter()
        orc.envelope_update(online, target, m, v, step[0], mk(), sw, n_actions=A, reward_dim=R, dedup=False)
        return time.perf_counter() - t0

    # torch's default (every hardware thread) is not the fastest setting for this update (BLAS calls of a few MFLOP each):
    # probe thread counts with one update each, walking DOWN from a quarter of the logical CPUs while it keeps improving (and
    # once up to a half), then time the rest of the sample at the best one.  The first call is the warm-up.
    all_threads = th.get_num_threads()
    one()
    probe = {}

    def probe_at(nt):
        th.set_num_threads(nt)
        probe[nt] = one()
        return probe[nt]

    nt = max(1, all_threads // 4)
    best_t = probe_at(nt)
    while nt > 1:                                   # downwards: 32 -> 16 -> 8 -> ... until it stops improving
        nxt = max(1, nt // 2)
        t = probe_at(nxt)
        if t >= best_t:
            break
        best_t, nt = t, nxt
    if max(1, all_threads // 2) not in probe and sum(probe.values()) < 0.6 * budget_s:
        probe_at(max(1, all_threads // 2))
    best = min(probe, key=probe.get)
    th.set_num_threads(best)
    timed = [probe[best]]
    t_start = time.perf_counter()
    while len(timed) < 3 and (time.perf_counter() - t_start) < 0.4 * budget_s:
        timed.append(one())
    sec = float(np.median(timed))
    # the same update with the next-state slabs evaluated once per (transition, weight) -- the B*W rows the GPU path computes --
    # instead of the reference's W^2*B: separates what the de-duplication buys from what the MI355X buys
    dd = []

    def one_dedup():
        sw = th.tensor(orc.random_weights(R, weights, "gaussian", rng=rng), dtype=th.float32)
        step[0] += 1
        t0 = time.perf_counter()
        orc.envelope_update(online, target, m, v, step[0], mk(), sw, n_actions=A, reward_dim=R, dedup=True)
        return time.perf_counter() - t0

    one_dedup()
    t_start = time.perf_counter()
    while len(dd) < 5 and (time.perf_counter() - t_start) < 0.15 * budget_s:
        dd.append(one_dedup())
    sec_dd = float(np.median(dd))
    th.set_num_threads(all_threads)
    return {"value": batch * weights / sec, "unit": "TD-updates/s", "cores": best, "threads_used": best,
            "host_logical_cpus": os.cpu_count(), "host_physical_cores": _physical_cores(),
            "threads_probed": {str(k): v for k, v in sorted(probe.items())},
            "kind": "port", "updates_per_s": 1.0 / sec,
            "dedup_value": batch * weights / sec_dd, "dedup_updates_per_s": 1.0 / sec_dd,
            "dedup_note": f"the same port with dedup=True (B*W-row targets, what the GPU path computes), {len(dd)} timed updates at "
                          f"{best} threads, median: the algorithmic share of the speed-up",
            "sample": f"{len(timed)} timed Envelope.update() steps (after a warm-up, at the best of {sorted(probe)} threads, probed "
                      f"downwards until it stopped improving) of the as-written reference algorithm (oracle/envelope_oracle.py, "
                      f"B={batch}, W={weights}, W^2*B-row targets) on torch-CPU, median"}


def _physical_cores():
    """Physical core count of this host (unique (package, core) pairs of /proc/cpuinfo); None if it cannot be read."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        return len(pairs) or None
    except OSError:
        return None


def committed_pmc():
    """The newest committed rocprofv3 PMC summary (profiles/r*_pmc_summary.json: FETCH_SIZE x2 + WRITE_SIZE per launch and kernel,
    collected as MI355X_MICROARCH.md prescribes by tools/pmc_summary.py on this same command).  PMC counters cannot be read from
    inside this process, so these figures are constants of the repository, not of this run; None if no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1]))["kernels"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def traffic_fields(bf16):
    """roofline.traffic = the DOMINANT kernel's own HBM bytes per launch (the chain kernel: mlp_chain_bf when the step ran on the
    bf16 matrix cores -- its forward launch --, else mlp_chain2); step_hbm_bytes = the sum over the step's kernels."""
    ks, src = committed_pmc()
    if not ks:
        return {"traffic": None, "traffic_source": None}
    name = "morl::mlp_chain_bf_kernel" if bf16 else "morl::mlp_chain2_kernel"
    if not any(k.startswith(name) for k in ks):          # (the committed summary is of the other arithmetic: say nothing rather than mix)
        return {"traffic": None, "traffic_source": f"{src} holds no {name} launches"}
    dom = next((v for k, v in ks.items() if k.startswith(name)), None)
    # per STEP: a kernel's per-launch average times its launches per step (the chain kernel runs twice: forward and backward-dX --
    # the figure of rounds 1-4 counted every kernel once and so left one chain launch out: 280 MB instead of 366 MB)
    steps = max((v.get("launches", 1) for k, v in ks.items() if "step_prologue" in k), default=0) or \
        min(v.get("launches", 1) for k, v in ks.items() if k.startswith("morl::") and "sumtree_set" not in k)
    step = sum(v["hbm_bytes"] * v.get("launches", steps) / steps for k, v in ks.items()
               if k.startswith("morl::") and "sumtree_set" not in k and "polyak" not in k)
    return {"traffic": dom["hbm_bytes"] if dom else None,
            "traffic_kernel": name if dom else None,
            "traffic_unit": "HBM bytes per launch of the dominant kernel (rocprofv3 PMC: FETCH_SIZE x 2 + WRITE_SIZE)",
            "traffic_source": f"committed profile {src} (PMC counters cannot be read in-process): a constant of the repository, "
                              "not of this run",
            "step_hbm_bytes": step, "algorithmic_bytes": ALGORITHMIC_BYTES_STEP,
            "step_hbm_over_algorithmic": step / ALGORITHMIC_BYTES_STEP,
            "step_hbm_note": "sum over the step's kernels of the same summary; SURVEY 8(d)'s 7.7 MB is the figure of a fully fused "
                             "forward + backward that keeps 117 MB of activations on chip -- here they are written once and read "
                             "once by the weight-gradient launch"}


def _sync(dev):
    if th.device(dev).type == "cuda":
        th.cuda.synchronize()


class _Stopwatch:
    """Elapsed milliseconds between two points of the device's stream (HIP events), or of the host clock on the emulated build
    (``--cpu-emulator``: launches execute synchronously there)."""

    def __init__(self, dev):
        self.cuda = th.device(dev).type == "cuda"
        if self.cuda:
            self.e0, self.e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)

    def start(self):
        if self.cuda:
            self.e0.record()
        else:
            self.t0 = time.perf_counter()

    def stop(self):
        if self.cuda:
            self.e1.record()
        else:
            self.t1 = time.perf_counter()

    def ms(self):
        return self.e0.elapsed_time(self.e1) if self.cuda else (self.t1 - self.t0) * 1e3


def _claim_stdout():
    """stdout must carry exactly ONE line, rank 0's JSON: C libraries (RCCL prints a version banner to stdout, flushed
    at exit, i.e. after the JSON) and the other ranks are moved to stderr; the result is written to the saved descriptor."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _self_spawn(n: int, argv, result_out) -> int:
    """``python bench.py --gpus N`` from a bare shell: re-execute under torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1) and pass rank 0's JSON line through to the real stdout."""
    import subprocess
    have = th.cuda.device_count()
    if have < n and "--shared-gpu" not in argv and "--cpu-emulator" not in argv:
        raise SystemExit(f"--gpus {n}: only {have} GPU(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] self-spawn:", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:                                   # the ranks keep stdout clean: only rank 0's JSON arrives here
        if ln.lstrip().startswith("{"):
            line = ln.strip()
        else:
            print(ln, end="", file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, file=result_out, flush=True)
    return rc if rc else (0 if line is not None else 1)


def _clock_ramp(dev, seconds=0.5):
    """The chip idles through the host-side set-up (agent construction, 20 000 buffer adds) and needs ~50 update steps
    (20 ms) to come back to its sustained clock -- longer than the driver's 5-step warm-up.  Half a second of dummy GEMMs
    before the warm-up steps brings it there (measured: first timed steps 0.44 -> 0.40 ms; tools/step_times.py).  Set-up,
    not workload: nothing of the timed region is pre-computed or cached by it."""
    x = th.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            x @ x
        th.cuda.synchronize()
    del x


def run_job(a, dist, world, rank, dev, W, sharded, steps, warmup, axis="batch", ramp=True, exact_f32=False):
    """Build the agent for a job of W sampled weights in total, run warm-up + `steps` timed Envelope.update() steps bracketed
    by barrier + synchronize on both sides; returns the measurements (wall = max over ranks).  exact_f32: every GEMM on the
    f32-input MFMA (morl_ctx_set_exact_f32: the "no split trick" leg of the line)."""
    from morl_baselines_amd.envelope import Envelope

    th.manual_seed(0)
    np.random.seed(0)
    B = a.batch
    agent = Envelope(SyntheticEnv(), learning_rate=3e-4, net_arch=ARCH, batch_size=B, gamma=0.99, max_grad_norm=1.0,
                     tau=1.0, target_net_update_freq=200, envelope=True, num_sample_w=W, per=bool(a.per),
                     per_alpha=0.6, buffer_size=100_000, gradient_updates=1, log=False, seed=0, device=dev,
                     engine=a.engine)
    if a.dw_mode is not None:
        agent.q_net.ctx.set_dw_mode(a.dw_mode)
    if exact_f32:
        agent.q_net.ctx.set_exact_f32(True)
    fill_buffer(agent.replay_buffer, a.buffer_fill, seed=0)
    agent.global_step = 1001
    if sharded:
        from morl_baselines_amd.distributed import shard_envelope_agent
        emu = (a.emulate_world, 0) if (a.emulate_world > 1 and world == 1) else None
        # batch axis: every rank runs the unsharded pipeline on B/N transitions, one all-reduce (the strong-scaled job);
        # weight axis: W/N weights per rank, all-gather of Q(w) + all-reduce (the weak-scaled job, whose weight axis grows)
        shard_envelope_agent(agent, dist, emulate=emu, axis=axis, transport=None if a.transport == "auto" else a.transport)
    transport = getattr(getattr(agent, "_shard", None), "transport", None)
    comm = getattr(getattr(agent, "_shard", None), "comm", None)
    comm_ranks = comm.size()[1] if comm is not None else None       # what the library's communicator itself reports (morl_comm_size)

    def step():
        agent.update()
        agent.global_step += 1

    if ramp and dev.type == "cuda":
        _clock_ramp(dev)
    for _ in range(max(warmup - 1, 0)):
        step()
    agent.q_net.ctx.set_timing(1)                    # the last warm-up step counts the chain launches of a step
    if warmup > 0:
        step()
    _sync(dev)
    kinds0 = agent.q_net.ctx.read_timing_kinds()
    launches_per_step = kinds0["forward"][0] + kinds0["forward2"][0] + kinds0["backward"][0]       # chain launches of one step
    if dist is not None:
        dist.barrier()
        _sync(dev)
    # chain launches are event-timed on the library's stream.  An event record costs ~3.5 us of stream time (four records
    # around the two chain launches of a step: +14 us = 4 %), so short runs (the driver's --steps 20) bracket ONE launch on
    # every second step, the launches of a step taking turns (a 5 + 20 run: 10 launches, 3 - 4 of each kind; bracketing one on
    # EVERY step cost the step ~5 us: 0.332 ms against 0.325 unbracketed), and long runs all launches of every 8th step.
    timing_every = int(os.environ.get("MORL_BENCH_TIMING", (-2 if steps >= 12 else -1) if steps <= 50 else 8))
    if not launches_per_step:
        timing_every = 1 if timing_every < 0 else timing_every
    agent.q_net.ctx.set_timing(timing_every)
    from morl_baselines_amd.ops import HostRing
    sw = _Stopwatch(dev)
    waited0 = HostRing.waited_s + agent.q_net.ctx.backpressure_seconds()
    t0 = time.perf_counter()
    sw.start()
    for _ in range(steps):
        step()
    sw.stop()
    t_enq = time.perf_counter() - t0                 # host time to enqueue the timed steps (no synchronisation inside) ...
    t_back = HostRing.waited_s + agent.q_net.ctx.backpressure_seconds() - waited0     # ... of which blocked on the device: a host that
    # enqueues faster than the device executes is throttled to the device's pace -- by the library (the target launch of a lazily
    # evaluated step is sized by the pair count of eight steps back) or a lap later by the pinned rings: back-pressure, not host work
    _sync(dev)
    if dist is not None:
        dist.barrier()
        _sync(dev)
    wall = time.perf_counter() - t0
    kinds = agent.q_net.ctx.read_timing_kinds()
    chain_kinds = ("forward", "forward2", "backward")
    n_chain, chain_ms = sum(kinds[k][0] for k in chain_kinds), sum(kinds[k][1] for k in chain_kinds)
    agent.q_net.ctx.set_timing(False)
    lazy_rows = agent.q_net.ctx.lazy_target_rows(agent.q_net.flat)      # distinct (b, j*) pairs of the LAST step (0: eager)
    gpu_ms = sw.ms()
    if dist is not None:
        t = th.tensor([wall], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=th.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    res = {"wall": wall, "host_enqueue_ms_per_step": (t_enq - t_back) * 1e3 / steps, "host_backpressure_ms_per_step": t_back * 1e3 / steps,
           "gpu_ms_per_step_events": gpu_ms / steps,
           "n_chain": n_chain, "chain_ms": chain_ms, "timed_steps": (len(range(0, steps, timing_every)) if timing_every > 0 else steps if timing_every == -1
                                                                  else (steps + 1) // 2 if timing_every == -2 else 0),
           "launches_per_step": launches_per_step, "timing_mode": timing_every, "kinds": kinds,
           "fwd_launches_per_step": kinds0["forward"][0], "fwd2_launches_per_step": kinds0["forward2"][0], "transport": transport, "comm_ranks": comm_ranks, "axis": axis if sharded else None,
           "lazy_target_rows": lazy_rows, "bf16": agent.q_net.ctx.last_step_bf16(),
           "loss": agent.last_loss(), "engine": agent.q_net.ctx.engine, "W": W, "B": B}
    del agent
    return res


def collective_microbench(dist, dev, world, n_allreduce, n_allgather, iters=20):
    """The step's two collectives alone, over every transport that comes up on this job -- RCCL inside the library and the
    single-hop transport over peer-mapped memory (morl_comm_ipc_*): microseconds per call of the all-reduce of
    [gradient | loss | priorities] and of the all-gather of Q(w), max over the ranks.  Every rank calls this; a transport
    that fails to come up anywhere is skipped by ALL ranks together, and the single-hop waits are bounded, so the worst case is
    a few seconds and an "error" entry -- never a hang."""
    from morl_baselines_amd.distributed import NativeComm
    from morl_baselines_amd.native import load_library
    lib = load_library()
    on_dev = dist.get_backend() == "nccl"
    side = dev if on_dev else th.device("cpu")
    out = {"allreduce_floats": n_allreduce, "allgather_floats_per_rank": n_allgather, "iters": iters}
    for name in (("rccl", "ipc") if on_dev else ("ipc",)):
        comm, err = None, None
        try:
            comm = NativeComm(lib, dist, dev, transport=name, max_allreduce=n_allreduce, max_allgather=n_allgather)
        except Exception as exc:                       # (the ipc set-up raises on every rank together; RCCL: agreed below)
            err = f"{type(exc).__name__}: {exc}"
        ok = th.tensor([0 if comm is None else 1], device=side)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            if comm is not None:
                comm.close()
            out[name] = {"error": err or "unavailable on another rank"}
            continue
        rec = {}
        buf = th.full((n_allreduce,), 1e-3, device=dev)
        send = th.full((n_allgather,), float(dist.get_rank()), device=dev)
        recv = th.zeros(world * n_allgather, device=dev)
        # every rank runs the SAME sequence of torch.distributed calls whatever fails locally: a failure is a flag that rides in
        # the all-reduce of the timing, and all ranks leave the loop together
        for what in ("allreduce", "allgather"):
            def call():
                if what == "allreduce":
                    comm.allreduce(buf)
                else:
                    comm.allgather_begin(send, recv)
                    comm.wait(recv)
            bad, us, why = 0.0, 0.0, None
            try:
                for _ in range(3):
                    call()
                _sync(dev)
            except Exception as exc:
                bad, why = 1.0, f"{type(exc).__name__}: {exc}"
            f = th.tensor([bad], dtype=th.float64, device=side)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)       # (also the barrier) nobody enters the timed loop unless everybody does
            bad = max(bad, float(f.item()))
            if not bad:
                try:
                    if what == "allreduce":
                        buf.fill_(1e-3)
                    sw = _Stopwatch(dev)
                    sw.start()
                    for _ in range(iters):
                        call()
                    sw.stop()
                    _sync(dev)
                    us = sw.ms() * 1e3 / iters
                    if name == "ipc":
                        comm.check()
                except Exception as exc:
                    bad, why = 1.0, f"{type(exc).__name__}: {exc}"
            t = th.tensor([us, bad], dtype=th.float64, device=side)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t[1].item()) > 0:
                rec["error"] = why or "failed on another rank"
                break
            rec[what + "_us"] = float(t[0].item())
        if "error" not in rec:
            rec["allgather_correct"] = bool((recv.view(world, -1)[:, 0].cpu() == th.arange(world, dtype=th.float32)).all())
        out[name] = rec
        dist.barrier()
        comm.close()
    return out


def _roofline(res, rows_rank):
    """Dominant kernel = the layer-fused chain (per step and rank: the launch with the online forward passes and the backward-dX
    launch): algorithmic flop of the timed launches over their summed HIP-event duration.  When the step ran on the bf16 matrix
    cores every fp32 product is six bf16 products: ``achieved`` / ``frac`` then price the SIX-fold flop against the dense bf16 peak,
    and ``fp32_equivalent_tflops`` is the algorithmic rate (what the same work would be called on the f32-input MFMA)."""
    n_chain, chain_ms, timed_steps = res["n_chain"], res["chain_ms"], res["timed_steps"]
    if chain_ms <= 0.0:          # (no usable event timings: the emulated build's events do not measure anything)
        n_chain = 0
    bf16 = bool(int(res.get("bf16") or 0) & 1)
    dw_bf16 = bool(int(res.get("bf16") or 0) & 2)
    # algorithmic flop of ONE launch of each kind (this rank's rows): the three-pass forward launch (or the launches of a sharded
    # step), the two-pass forward launch of a lazily evaluated step, the backward-dX launch, the weight-gradient launch
    flop_kind = {"forward": rows_rank * 3 * FWD_FLOP_ROW / max(1, res.get("fwd_launches_per_step") or 1),
                 "forward2": rows_rank * 2 * FWD_FLOP_ROW / max(1, res.get("fwd2_launches_per_step") or 1), "backward": rows_rank * BWD_DX_FLOP_ROW, "dw": rows_rank * FWD_FLOP_ROW}
    kinds = res.get("kinds") or {}
    chain_flop = sum(flop_kind[k] * kinds[k][0] for k in ("forward", "forward2", "backward") if k in kinds)
    launches_per_step = res.get("launches_per_step") or (n_chain / timed_steps if timed_steps else 0)
    flop_per_launch = chain_flop / n_chain if n_chain else float("nan")
    avg_launch_s = (chain_ms * 1e-3 / n_chain) if n_chain else float("nan")
    fp32_eq = chain_flop / (chain_ms * 1e-3) / 1e12 if n_chain else float("nan")
    mult, peak = (BF16_PRODUCTS, PEAK_BF16_MFMA_TFLOPS) if bf16 else (1, PEAK_FP32_MFMA_TFLOPS)
    achieved = fp32_eq * mult
    # the three GEMM kernels of the step one by one (the launches of a kind that were bracketed; algorithmic flop of this rank's
    # rows over their mean duration)
    per_kernel = {}
    chain_name = "mlp_chain_bf (split-bf16, 6 products)" if bf16 else "mlp_chain2 (f32-input MFMA)"
    name_kind = {"forward": chain_name + " forward (3 passes)", "forward2": chain_name + " forward (2 passes: online next-state + training)",
                 "backward": chain_name + " backward-dX",
                 "dw": "dw_bf (dW, db; split-bf16, 6 products)" if dw_bf16 else "dw_tiles (dW, db; f32-input MFMA)"}
    for k, (n_k, ms_k) in kinds.items():
        if n_k and ms_k > 0.0:
            us = ms_k * 1e3 / n_k
            tf = flop_kind[k] / (us * 1e-6) / 1e12
            on_bf = dw_bf16 if k == "dw" else bf16
            per_kernel[k] = {"kernel": name_kind[k], "launches_timed": n_k, "avg_launch_us": us,
                             "algorithmic_flop_per_launch": flop_kind[k], "fp32_equivalent_tflops": tf,
                             "achieved": tf * (BF16_PRODUCTS if on_bf else 1), "peak": PEAK_BF16_MFMA_TFLOPS if on_bf else PEAK_FP32_MFMA_TFLOPS,
                             "frac": tf * (BF16_PRODUCTS if on_bf else 1) / (PEAK_BF16_MFMA_TFLOPS if on_bf else PEAK_FP32_MFMA_TFLOPS)}
    out = {"bound": "mfma",
           "kernel": ("mlp_chain_bf (layer-fused Q-net forward / backward-dX, six split-bf16 products per fp32 product on "
                      "v_mfma_f32_16x16x32_bf16)" if bf16 else "mlp_chain2 (layer-fused Q-net forward / backward-dX, f32-input MFMA)"),
           "per_kernel": per_kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
           "peak_source": ("MI355X_MICROARCH.md: dense bf16 MFMA 2.5 PFLOP/s (no sparsity); executed flop = 6 x algorithmic" if bf16
                           else "MI355X_MICROARCH.md: f32-input MFMA 157.3 TFLOP/s"),
           "fp32_equivalent_tflops": fp32_eq, "frac_of_fp32_mfma_peak": fp32_eq / PEAK_FP32_MFMA_TFLOPS,
           "launches_timed": n_chain, "timed_steps": timed_steps, "launches_per_step": launches_per_step,
           "timing": ("one launch per step, taking turns" if res.get("timing_mode") == -1 else
                      "one launch of every second step, taking turns" if res.get("timing_mode") == -2 else
                      "all launches of every %d-th step" % res.get("timing_mode", 0)),
           "avg_launch_us": avg_launch_s * 1e6,
           "algorithmic_flop_per_launch": flop_per_launch}
    out.update(traffic_fields(bf16))
    return out


def whole_step_bf16_frac(rows, ms, bf16):
    """The step's EXECUTED matrix-core work over the dense bf16 peak: six bf16 products per fp32 product of the launches that ran on
    the bf16 pipe (bit 0 of `bf16`: the two online forward passes and the dX backward; bit 1: the weight gradients too) over the
    whole step's time, every non-GEMM microsecond included.  None for a step on the f32-input MFMA."""
    if not (int(bf16 or 0) & 1) or ms <= 0.0:
        return None
    flop = rows * (2 * FWD_FLOP_ROW + BWD_DX_FLOP_ROW + (FWD_FLOP_ROW if int(bf16) & 2 else 0))
    return BF16_PRODUCTS * flop / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS


def one_step_parity(dev, case=None, lib=None):
    """bench.py checks what it times: ONE gradient step at the metric's full shape (256 x 64 x 3, the seeded inputs of the reference
    fixture tests/golden/envelope_flagship_full.npz) through the library's DEFAULT pipeline (lazy targets, bf16 matrix cores when
    enabled, the PER tree update inside the step) against the oracle on the host:
      * loss and gradient norm to 1e-5;
      * the B priorities |td . w| within the tolerance the 1e-5 contract on Q implies ((1 + gamma) * 1e-5 * max|Q|: w is
        L1-normalised), the PER sum tree's root equal to what the oracle's tree (prioritized_buffer.py:69-82, 187-195) makes of
        the device's own priorities (to 3e-7: the device's powf may differ from numpy's fp32 power by an ulp);
      * the envelope arg-max of all 16 384 TD rows (a second step from the same state that asks for the indices): every row that
        differs from the oracle's must be a near-tie -- runner-up within 4e-7 of the maximum --, and no more rows may differ than
        have such a runner-up (counted from the oracle's own slab).
    Part of the cpu_baseline leg (the only place the benchmark may touch oracle/); a pipeline that disagrees prints no line."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import envelope_oracle as orc
    from cases import FLAGSHIP, make_inputs
    import morl_baselines_amd.ops as ops
    c = case or FLAGSHIP                  # (tests/test_bench_line.py runs the same checks on a small fixture through the emulator)
    inp = make_inputs(c)
    flat = lambda ps: th.cat([th.as_tensor(p).reshape(-1) for p in ps])
    ctx = ops.QNetContext(c.D, c.R, c.A, c.arch, c.B, c.W, lib=lib)
    lib = ctx.lib
    batch = (th.tensor(inp["obs"]).to(dev), th.tensor(inp["next_obs"]).to(dev),
             th.tensor(inp["actions"].astype(np.int32).reshape(-1)).to(dev), th.tensor(inp["rewards"]).to(dev),
             th.tensor(inp["dones"]).reshape(-1).to(dev), th.tensor(inp["sampled_w"]).float().to(dev))

    def state():
        po, pt = flat(inp["online"]).to(dev), flat(inp["target"]).to(dev)
        return po, pt, th.zeros_like(po), flat(inp["exp_avg"]).to(dev), flat(inp["exp_avg_sq"]).to(dev)

    # (1) the default pipeline, the step's PER update included: a tree of B leaves at the running maximum, transition b <-> leaf b
    RMAX, ALPHA = 0.125, 0.6
    n_levels = int(np.ceil(np.log2(c.B))) + 1
    tree = th.zeros(2 ** n_levels - 1, dtype=th.float64, device=dev)
    rmax = th.tensor([RMAX], dtype=th.float64, device=dev)
    idx = th.arange(c.B, dtype=th.int64, device=dev)
    ops.sumtree_set(lib, tree, n_levels, idx, None, rmax)
    po, pt, g, m, v = state()
    res = ops.envelope_update(ctx, po, pt, g, m, v, *batch, gamma=c.gamma, lr=c.lr, adam_step=c.step, max_grad_norm=c.max_grad_norm,
                              per=(tree, n_levels, idx, ALPHA, rmax))
    lazy_rows, bf16 = ctx.lazy_target_rows(po), ctx.last_step_bf16()
    # (2) the same step again from the same state, asking for the arg-max indices (still lazily evaluated)
    po2, pt2, g2, m2, v2 = state()
    res2 = ops.envelope_update(ctx, po2, pt2, g2, m2, v2, *batch, gamma=c.gamma, lr=c.lr, adam_step=c.step,
                               max_grad_norm=c.max_grad_norm, debug="lazy")
    if th.device(dev).type == "cuda":
        th.cuda.synchronize()
    nt = th.get_num_threads()
    th.set_num_threads(max(1, min(32, (os.cpu_count() or 4) // 4)))
    sw = th.tensor(inp["sampled_w"]).float()
    o = orc.envelope_update([th.tensor(a) for a in inp["online"]], [th.tensor(a) for a in inp["target"]],
                            [th.tensor(a) for a in inp["exp_avg"]], [th.tensor(a) for a in inp["exp_avg_sq"]], c.step,
                            tuple(th.tensor(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones")),
                            sw, n_actions=c.A, reward_dim=c.R, gamma=c.gamma, lr=c.lr,
                            max_grad_norm=c.max_grad_norm, dedup=True, apply_step=False)
    th.set_num_threads(nt)
    rel_l = abs(res["loss"].item() - o["loss"].item()) / abs(o["loss"].item())
    rel_g = abs(res["grad_norm"].item() - o["grad_norm"].item()) / abs(o["grad_norm"].item())
    # priorities and the tree
    raw_dev = res["priority"].cpu().numpy()
    raw_ref = o["priority_raw"].numpy().astype(np.float64)
    qmax = max(1.0, float(o["q_values"].abs().max()), float(o["target"].abs().max()))
    prio_tol = (1.0 + c.gamma) * 1e-5 * qmax
    prio_err = float(np.abs(raw_dev.astype(np.float64) - raw_ref).max())
    host_tree = orc.SumTree(c.B)
    host_tree.batch_set(np.arange(c.B), np.full(c.B, RMAX))
    pr = (raw_dev + np.float32(RMAX)) ** np.float32(ALPHA)                 # envelope.py:333 in numpy's fp32, as the reference does
    host_tree.batch_set(np.arange(c.B), pr.astype(np.float64))
    root_dev, root_host = float(tree[0].item()), float(host_tree.nodes[0][0])
    root_rel = abs(root_dev - root_host) / root_host
    # arg-max rows
    pref_d, ac_d = res2["pref"].cpu().long(), res2["ac"].cpu().long()
    mism = ((pref_d != o["pref"].reshape(-1).long()) | (ac_d != o["ac"].reshape(-1).long())).nonzero().flatten()
    s_all = th.einsum("ir,bjar->ibja", sw.double(), o["qo"].double().view(c.B, c.W, c.A, c.R)).reshape(c.W, c.B, c.W * c.A)
    top2 = s_all.topk(2, dim=-1).values
    at_risk = (((top2[..., 0] - top2[..., 1]) / top2[..., 0].abs().clamp(min=1.0)) <= 4e-7).reshape(-1)
    flips_ok = mism.numel() <= int(at_risk.sum()) and bool(at_risk[mism].all())
    ctx.close()
    out = {"shape": f"B={c.B} x W={c.W} x R={c.R} (fixture inputs of tests/golden/envelope_{c.name}.npz)", "loss_hip": res["loss"].item(),
           "loss_oracle": o["loss"].item(), "loss_rel": rel_l, "grad_norm_rel": rel_g, "lazy_target_rows": lazy_rows, "bf16_matrix_cores": bf16,
           "tolerance": 1e-5,
           "priority_max_abs_err": prio_err, "priority_tolerance": prio_tol,
           "per_tree_root": root_dev, "per_tree_root_rel_err_given_device_priorities": root_rel,
           "argmax_rows": c.B * c.W, "argmax_rows_differing": int(mism.numel()), "argmax_rows_with_a_near_tie": int(at_risk.sum()),
           "argmax_near_tie": 4e-7,
           "ok": bool(rel_l <= 1e-5 and rel_g <= 1e-5 and prio_err <= prio_tol and root_rel <= 3e-7 and flips_ok)}
    if not out["ok"]:
        raise SystemExit(f"bench.py: the timed pipeline disagrees with the oracle on one step: {out}")
    return out


def main():
    result_out = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--per", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--weights", type=int, default=64)
    ap.add_argument("--engine", type=int, default=None)
    ap.add_argument("--dw-mode", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 headline: strong = the metric's fixed 256 x 64 x 3 update, its 64 weights split over the "
                         "GPUs (default); weak = every GPU keeps 64 weights (the job's weight axis grows to 64*N and every "
                         "TD row's envelope max runs over all 64*N candidates after the all-gather).  The other one is "
                         "measured too and attached as a labelled sub-record unless --no-sub-record")
    ap.add_argument("--no-sub-record", action="store_true")
    ap.add_argument("--no-ramp-record", action="store_true",
                    help="skip the extra un-ramped run of the single-GPU job (ms_per_step_no_ramp)")
    ap.add_argument("--no-sustained-record", action="store_true",
                    help="skip the long run of the single-GPU job (ms_per_step_sustained: >= 2 000 steps, the lazily selected row "
                         "count at its long-run level)")
    ap.add_argument("--sustained-steps", type=int, default=2000)
    ap.add_argument("--no-exact-record", action="store_true",
                    help="skip the run of the single-GPU job with every GEMM on the f32-input MFMA (exact_f32_ms_per_step)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="with --force-shard on one GPU: run the step of rank 0 of a job of this many ranks (its kernels, "
                         "launches, host work and message sizes; the other ranks' slabs are zeros) -- a measurement aid, "
                         "the line it prints is labelled as such")
    ap.add_argument("--shard-axis", choices=["auto", "batch", "weights"], default="auto",
                    help="how N > 1 ranks split one update: batch (B/N transitions per rank, one all-reduce), weights (W/N "
                         "weights per rank: all-gather of Q(w) + all-reduce, the north_star's description); auto = the "
                         "strong-scaled job is measured on BOTH and the faster one is the headline, the weak-scaled one runs "
                         "on the weight axis (the one that grows)")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="FUNCTIONAL check of the N > 1 code path on a one-GPU box: the N ranks all run on cuda:0 (gloo process "
                         "group; collectives over the single-hop hipIpc transport, MORL_COMM=ipc, unless MORL_COMM says "
                         "otherwise -- RCCL refuses duplicate devices).  The line is labelled: its timings describe N processes "
                         "sharing one chip, not a multi-GPU job")
    ap.add_argument("--no-collective-microbench", action="store_true",
                    help="N > 1: skip the side record that times the step's two collectives alone over RCCL and over the "
                         "single-hop transport (bounded: a few seconds at worst)")
    ap.add_argument("--transport", choices=["auto", "rccl", "ipc", "torch", "staged"], default="auto",
                    help="collectives of the N > 1 rank steps: auto = MORL_COMM, else RCCL inside libmorl_hip.so on an RCCL process group "
                         "(falling back, on every rank together, to the torch.distributed call-backs); staged = the library calls of a "
                         "rank step one by one with torch.distributed between them")
    ap.add_argument("--buffer-fill", type=int, default=20_000, help="seeded transitions in the replay buffer (BASELINE.md s3: 20 000)")
    ap.add_argument("--cpu-emulator", action="store_true",
                    help="FUNCTIONAL check without a GPU (the CPU test-suite's end-to-end run of this script): the ranks run the "
                         "host-emulated kernel library (MORL_HIP_LIB must point at tests/hipsim's build) on CPU tensors over gloo. "
                         "Its timings mean nothing and the line says so; never the product path")
    ap.add_argument("--force-shard", action="store_true",
                    help="run the weight-sharded step (RCCL collectives) even with one rank (path check on a 1-GPU box)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(_self_spawn(a.gpus, sys.argv[1:], result_out))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.cpu_emulator:
        from morl_baselines_amd.native import load_library
        if load_library().is_device_build:
            raise SystemExit("--cpu-emulator needs MORL_HIP_LIB to point at the host-emulated test build (tests/hipsim)")
        th.set_num_threads(1)
        dev = th.device("cpu")
    else:
        if not th.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
        if a.shared_gpu:
            local_rank = 0                                    # every rank on the one GPU of the box
            os.environ.setdefault("MORL_COMM", "ipc")
        th.cuda.set_device(local_rank)
        dev = th.device("cuda", local_rank)
    dist = None
    if world > 1 or a.force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.shared_gpu or a.cpu_emulator:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        print(f"[bench] rank {rank}/{world} on {dev}: {dist.get_backend()} nranks={dist.get_world_size()}", file=sys.stderr, flush=True)
    sharded = dist is not None

    if a.weights % world:
        raise SystemExit(f"--weights {a.weights} must be divisible by the number of ranks ({world})")
    B = a.batch
    parts = max(world, a.emulate_world if a.force_shard else 1)
    if sharded and B % parts:
        raise SystemExit(f"--batch {B} must be divisible by the number of ranks")

    def job(W, axis, ramp=True, steps=None, exact_f32=False):
        """One measured job; every rank must reach the same verdict, so a failure is agreed on through an all-reduce."""
        try:
            res = run_job(a, dist, world, rank, dev, W, sharded, steps or a.steps, a.warmup, axis, ramp, exact_f32)
        except Exception as exc:
            import traceback
            traceback.print_exc()
            res = {"error": f"{type(exc).__name__}: {exc}"}
        if dist is not None and world > 1:
            bad = th.tensor([1 if "error" in res else 0], device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()) and "error" not in res:
                res = {"error": "another rank failed"}
        return res

    # N > 1: the STRONG-scaled job (the metric's fixed 256 x 64 x 3 update) is measured on BOTH partitions -- the weight axis
    # BASELINE.json's north_star describes (W/N weights per rank: all-gather of Q(w) + all-reduce) and the batch axis (B/N
    # transitions per rank: one all-reduce) -- and both figures go into the line; the headline is the faster one and
    # config.shard_axis says which.  The WEAK-scaled job (64 weights per GPU, weight axis) is the labelled sub-record.
    strong, weak, no_ramp, sustained, exact = {}, None, None, None, None
    if sharded:
        axes = ["batch", "weights"] if a.shard_axis == "auto" else [a.shard_axis]
        if a.scaling == "weak":
            weak = job(a.weights * world, "weights" if a.shard_axis == "auto" else a.shard_axis)
            if not a.no_sub_record:
                for ax in axes:
                    strong[ax] = job(a.weights, ax)
        else:
            for ax in axes:
                strong[ax] = job(a.weights, ax)
            if world > 1 and not a.no_sub_record:
                weak = job(a.weights * world, "weights" if a.shard_axis == "auto" else a.shard_axis)
    else:
        if not a.no_ramp_record:
            # the same run shape WITHOUT the clock ramp, measured first (the chip still idles from the set-up): BASELINE.md s3's
            # procedure as written; the headline below runs behind the ramp and says so in config.setup
            no_ramp = job(a.weights, None, ramp=False)
        strong["single"] = job(a.weights, None)
        if not a.no_sustained_record:
            # the same job over a long run: the lazily selected row count grows over the first ~ 1 000 updates of a fresh agent
            # (1 193 -> 1 749 of 16 384 at the flagship shape), so a 20-step line flatters the target rows' launch
            sustained = job(a.weights, None, steps=max(a.sustained_steps, a.steps))
        if not a.no_exact_record:
            exact = job(a.weights, None, exact_f32=True)        # what the step costs without the split-bf16 products

    coll = None
    if world > 1 and not a.no_collective_microbench:
        # the A/B SURVEY 8(e) asks for: the same two messages over RCCL's ring / tree and over single-hop direct writes
        n_params = 36 * 256 + 3 * 257 * 256 + 257 * A * R       # P of the flagship net
        try:
            # (the weight-sharded step's all-gather: both slabs, or the online one alone when it evaluates its targets lazily)
            w_res = strong.get("weights") or weak or {}
            nets = 1 if (w_res.get("lazy_target_rows") or 0) > 0 else 2
            coll = collective_microbench(dist, dev, world, n_params + 1 + B, nets * B * (a.weights // world) * A * R)
        except Exception as exc:
            coll = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        def record(res, W, scaling):
            if "error" in res:
                return dict(res)
            rows_step = B * W                             # TD rows per gradient step of the whole job
            ms = res["wall"] * 1e3 / a.steps
            # (an EMULATED record times one rank of the job run alone: everything below then describes that rank's rows, not the job's)
            emulated = a.force_shard and a.emulate_world > 1 and world == 1
            rows_timed = rows_step // parts if emulated else rows_step
            lazy_all = (1 if emulated else parts) * (res.get("lazy_target_rows") or 0)
            return {"value": rows_timed * a.steps / res["wall"], "unit": "TD-updates/s", "ms_per_step": ms,
                    "scaling": scaling, "weights": W, "weights_per_gpu": W // world, "shard_axis": res.get("axis"),
                    "transport": res.get("transport"),
                    "updates_per_s": a.steps / res["wall"],
                    "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"],
                    "host_backpressure_ms_per_step": res.get("host_backpressure_ms_per_step", 0.0),
                    "gpu_ms_per_step_events": res["gpu_ms_per_step_events"], "last_loss": res["loss"],
                    "roofline": _roofline(res, rows_step // parts),
                    "lazy_target_rows_last_step": res.get("lazy_target_rows"),
                    "bf16": int(res.get("bf16") or 0),
                    "rows_timed": rows_timed,
                    "whole_step_algorithmic_tflops": rows_timed * (4 * FWD_FLOP_ROW + BWD_DX_FLOP_ROW) / (ms * 1e-3) / 1e12,
                    "whole_step_executed_tflops": ((rows_timed * (3 * FWD_FLOP_ROW + BWD_DX_FLOP_ROW) + lazy_all * FWD_FLOP_ROW)
                                                   if res.get("lazy_target_rows")
                                                   else rows_timed * (4 * FWD_FLOP_ROW + BWD_DX_FLOP_ROW)) / (ms * 1e-3) / 1e12}

        single = world == 1 and not a.force_shard
        scaling = "single" if world == 1 else a.scaling        # (one GPU: neither weak nor strong -- nothing is split)
        if a.scaling == "weak" and world > 1:
            head_res, W_head = weak, a.weights * world
        else:
            ok = {ax: r for ax, r in strong.items() if "error" not in r}
            if not ok:
                raise SystemExit(f"every job failed: {strong}")
            head_axis = min(ok, key=lambda ax: ok[ax]["wall"])
            head_res, W_head = ok[head_axis], a.weights
        if "error" in head_res:
            raise SystemExit(f"the headline job failed: {head_res}")
        h = record(head_res, W_head, scaling)
        head_axis = head_res.get("axis")
        out = {
            "metric": "Envelope-Q TD updates/sec (batch x weights x obj = 256 x 64 x 3)",
            "value": h["value"],
            "unit": "TD-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": h["ms_per_step"],
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": ("f32 (6 x bf16 split products, fp32 accumulate: online forward passes, dX backward" +
                      (", dW; f32-input MFMA: target rows)" if h["bf16"] & 2 else "; f32-input MFMA: target rows, dW)") if h["bf16"] & 1 else "f32"),
            "data": "synthetic",
            "config": {"workload": f"Envelope.update(): B={B} x W={W_head} x R={R}, obs {D}, {A} actions, net {ARCH}, "
                                   f"PER {'on' if a.per else 'off'}, buffer 20k seeded transitions (BASELINE.md s3)",
                       "global_batch": B, "weights": W_head, "objectives": R,
                       "weights_per_gpu": W_head // world,
                       "parallelism": "single GPU" if single else (
                           f"batch axis sharded over {parts} ranks, {B // parts} transitions x {W_head} weights each "
                           f"({scaling} scaling; one all-reduce of gradient | loss | priorities)" if head_axis == "batch" else
                           f"weight axis sharded over {parts} ranks, {W_head // parts} weights each ({scaling} scaling; "
                           "all-gather of Q(w), all-reduce of gradient | loss | priorities)"),
                       "shard_axis": head_axis,
                       "transport": head_res.get("transport"),
                       # ranks the step's communicator ran over, as the library reports them (morl_comm_size); rccl_ranks: the same
                       # when that communicator is RCCL inside libmorl_hip.so (ncclCommCount), null for every other transport
                       "comm_ranks": head_res.get("comm_ranks"),
                       "rccl_ranks": head_res.get("comm_ranks") if str(head_res.get("transport") or "").startswith("rccl") else None,
                       "engine": head_res["engine"],
                       "arithmetic": ("online forward passes, dX backward" + (" and weight gradients" if h["bf16"] & 2 else "") +
                                      " on the bf16 matrix cores as six split-bf16 products per fp32 product (csrc/mlp_chain_bf.h, "
                                      "dw_bf.h: fp32-class accuracy); MORL_EXACT_F32=1 keeps every GEMM on the f32-input MFMA"
                                      if h["bf16"] & 1 else "every GEMM on the f32-input MFMA (exact fp32 fma chains)"),
                       "setup": "0.5 s device clock ramp (dummy GEMMs) before the warm-up steps; ms_per_step_no_ramp is the same "
                                "run shape without it"},
            "updates_per_s": h["updates_per_s"],
            "scalar_td_per_s": h["value"] * R,
            "gpu_ms_per_step_events": h["gpu_ms_per_step_events"],
            "host_enqueue_ms_per_step": h["host_enqueue_ms_per_step"],
            "host_backpressure_ms_per_step": h["host_backpressure_ms_per_step"],
            "host_enqueue_note": "host time to enqueue a step (Python + ctypes + launches), the time blocked on the device excluded: a host "
                                 "that runs ahead waits in the pinned rings a lap later (host_backpressure_ms_per_step)",
            "last_loss": h["last_loss"],
            # whole-step fractions are fp32-EQUIVALENT rates over the f32-input MFMA peak (the one common yardstick of a step that
            # mixes both instruction families); algorithmic = SURVEY 8(d)'s five full passes, executed = what ran (lazy targets)
            "roofline": dict(h["roofline"], frac_whole_step_bf16=whole_step_bf16_frac(h["rows_timed"], h["ms_per_step"], h["bf16"]),
                             whole_step_algorithmic_tflops=h["whole_step_algorithmic_tflops"],
                             whole_step_executed_tflops=h["whole_step_executed_tflops"],
                             whole_step_frac_algorithmic=h["whole_step_algorithmic_tflops"] / PEAK_FP32_MFMA_TFLOPS,
                             whole_step_frac_executed=h["whole_step_executed_tflops"] / PEAK_FP32_MFMA_TFLOPS),
            "lazy_target_rows_last_step": h["lazy_target_rows_last_step"],
        }
        if h["lazy_target_rows_last_step"]:
            out["config"]["target_evaluation"] = (
                f"lazy: the target network is evaluated after the arg-max, on the {h['lazy_target_rows_last_step']} distinct "
                f"(transition, weight) pairs the TD rows of the last step selected (of {B * W_head // parts} per rank) -- same values "
                "as the full target slab; whole_step_algorithmic_tflops counts SURVEY 8(d)'s five full passes, "
                "whole_step_executed_tflops what ran; MORL_LAZY_TARGETS=0 evaluates the slab eagerly")
        if no_ramp is not None and "error" not in no_ramp:
            out["ms_per_step_no_ramp"] = no_ramp["wall"] * 1e3 / a.steps
        if sustained is not None and "error" not in sustained:
            n_s = max(a.sustained_steps, a.steps)
            out["ms_per_step_sustained"] = sustained["wall"] * 1e3 / n_s
            out["sustained"] = {"steps": n_s, "lazy_target_rows_last_step": sustained.get("lazy_target_rows"),
                                "value": B * W_head * n_s / sustained["wall"],
                                "frac_whole_step_bf16": whole_step_bf16_frac(B * W_head, sustained["wall"] * 1e3 / n_s, sustained.get("bf16")),
                                "note": "the headline job over a long run (same agent set-up, warm-up and clock ramp): the lazily selected "
                                        "target-row count at its long-run level"}
        if exact is not None and "error" not in exact:
            out["exact_f32_ms_per_step"] = exact["wall"] * 1e3 / a.steps
            out["exact_f32"] = {"bf16": int(exact.get("bf16") or 0), "value": B * W_head * a.steps / exact["wall"],
                                "note": "the headline job with every GEMM on the f32-input MFMA (morl_ctx_set_exact_f32 / MORL_EXACT_F32=1): "
                                        "the cost of not using the six split-bf16 products"}
        if world > 1 or a.force_shard:
            # both partitions of the strong-scaled job, each labelled; the headline above is the faster one
            out["strong_scaling_axes"] = {
                ax: dict(record(r, a.weights, "strong"),
                         note=("north_star's partition: W/N weights per rank, all-gather of Q(w) + all-reduce" if ax == "weights"
                               else "B/N transitions per rank, all W weights: one all-reduce, no all-gather"))
                for ax, r in strong.items()}
            # what one rank of an N-rank job takes when run ALONE (bench.py --force-shard --emulate-world N, profiles/): the
            # ceiling of strong scaling before any collective costs a microsecond -- nobody should read >= 6x into this record
            out["config"]["strong_scaling_ceiling_emulated"] = emulated_ceiling()
        if coll is not None:
            out["collectives_alone"] = dict(coll, note="microseconds per call, max over the ranks, HIP events around 20 back-to-back "
                                                       "calls; rccl = RCCL inside libmorl_hip.so, ipc = single-hop direct writes over "
                                                       "peer-mapped memory (MORL_COMM=ipc selects it for the step)")
        if a.cpu_emulator:
            out["cpu_emulator"] = ("FUNCTIONAL record, not a measurement: the ranks ran the host-emulated kernel library on CPU tensors "
                                   "(gloo); every timing in this line describes the emulator")
        if a.shared_gpu and world > 1:
            out["shared_gpu"] = (f"FUNCTIONAL record, not a multi-GPU measurement: the {world} ranks of this job shared ONE MI355X "
                                 "(gloo process group); value / ms_per_step describe that")
        if a.force_shard and a.emulate_world > 1 and world == 1:
            share = (f"{B // a.emulate_world} transitions x {W_head} weights" if head_axis == "batch"
                     else f"{B} transitions x {W_head // a.emulate_world} weights")
            out["emulated"] = (f"NOT a job throughput: the step of rank 0 of a {a.emulate_world}-rank job ({share}) run alone on "
                               "one GPU; value / ms_per_step / the whole_step_* rates describe that rank's rows")
        if world > 1 and a.scaling == "strong" and weak is not None:
            out["weak_scaling"] = dict(record(weak, a.weights * world, "weak"),
                                       note=f"sub-record, NOT the headline: weak scaling, W = {a.weights * world} sampled weights in "
                                            f"total ({a.weights} per GPU; weight axis sharded), same steps / warm-up")
        if not a.no_cpu_baseline and world == 1:
            out["one_step_parity"] = one_step_parity(dev)
            cb = cpu_baseline(B, W_head, bool(a.per))
            out["cpu_baseline"] = cb
            total = out["value"] / cb["value"]
            algo = cb["dedup_value"] / cb["value"]
            out["speedup_vs_cpu_baseline"] = total
            out["speedup_factors"] = {"total": total, "algorithmic_dedup": algo, "hardware": total / algo,
                                      "note": "total = GPU / CPU-as-written; algorithmic_dedup = CPU(B*W-row targets) / "
                                              "CPU(as written, W^2*B rows); hardware = GPU / CPU(B*W-row targets)"}
        def finite(x):           # (strict JSON: a figure that could not be measured is null, not NaN)
            if isinstance(x, float) and x != x:
                return None
            if isinstance(x, dict):
                return {k: finite(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [finite(v) for v in x]
            return x
        print(json.dumps(finite(out)), file=result_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
