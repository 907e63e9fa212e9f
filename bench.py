"""Envelope-Q TD-update benchmark (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--per 0|1] [--weights 64] [--batch 256]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` from a bare shell (no WORLD_SIZE in the environment) re-executes itself under
``torch.distributed.run`` with N ranks on 127.0.0.1, one per GPU, and passes rank 0's JSON line through.
N > 1: the headline is STRONG scaling of the metric's fixed 256 x 64 x 3 update (the 64 sampled weights are split over the
ranks, ``config.weights`` stays 64); the weak-scaled job (64 weights per GPU, W = 64*N) is measured right after it and
reported as the clearly labelled sub-record ``weak_scaling`` of the same line (``--scaling weak`` makes it the headline).

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

    # torch's default (every hardware thread) is not necessarily the fastest setting: probe a few counts with one update
    # each (the first call also serves as the warm-up), then time the rest of the sample at the best one
    all_threads = th.get_num_threads()
    cands = sorted({max(1, all_threads // 4), max(1, all_threads // 2), all_threads})
    one()
    probe = {}
    for nt in cands:
        th.set_num_threads(nt)
        probe[nt] = one()
    best = min(probe, key=probe.get)
    th.set_num_threads(best)
    timed = [probe[best]]
    t_start = time.perf_counter()
    while len(timed) < 3 and (time.perf_counter() - t_start) < budget_s:
        timed.append(one())
    th.set_num_threads(all_threads)
    sec = float(np.median(timed))
    return {"value": batch * weights / sec, "unit": "TD-updates/s", "cores": best, "threads_used": best,
            "host_logical_cpus": os.cpu_count(), "host_physical_cores": _physical_cores(),
            "threads_probed": {str(k): v for k, v in probe.items()},
            "kind": "port", "updates_per_s": 1.0 / sec,
            "sample": f"{len(timed)} timed Envelope.update() steps (after a warm-up, at the best of {cands} threads) of the "
                      f"as-written reference algorithm (oracle/envelope_oracle.py, B={batch}, W={weights}, W^2*B-row targets) "
                      "on torch-CPU, median"}


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


def measured_chain_traffic():
    """HBM bytes per mlp_chain launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected as
    MI355X_MICROARCH.md prescribes; profiles/*_pmc_summary.json).  PMC counters cannot be read from inside this
    process, so the figure is the one measured offline on this same command; None if no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        sel = [v for k, v in ks.items() if k.startswith("morl::mlp_chain")]
        n = sum(v["launches"] for v in sel)
        return sum(v["hbm_bytes"] * v["launches"] for v in sel) / n if n else None
    except Exception:
        return None


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
    if have < n:
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


def run_job(a, dist, world, rank, dev, W, sharded, steps, warmup, axis="batch"):
    """Build the agent for a job of W sampled weights in total, run warm-up + `steps` timed Envelope.update() steps bracketed
    by barrier + synchronize on both sides; returns the measurements (wall = max over ranks)."""
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
    fill_buffer(agent.replay_buffer, 20_000, seed=0)
    agent.global_step = 1001
    if sharded:
        from morl_baselines_amd.distributed import shard_envelope_agent
        emu = (a.emulate_world, 0) if (a.emulate_world > 1 and world == 1) else None
        # batch axis: every rank runs the unsharded pipeline on B/N transitions, one all-reduce (the strong-scaled job);
        # weight axis: W/N weights per rank, all-gather of Q(w) + all-reduce (the weak-scaled job, whose weight axis grows)
        shard_envelope_agent(agent, dist, emulate=emu, axis=axis)

    def step():
        agent.update()
        agent.global_step += 1

    _clock_ramp(dev)
    for _ in range(max(warmup - 1, 0)):
        step()
    agent.q_net.ctx.set_timing(1)                    # the last warm-up step counts the chain launches of a step
    if warmup > 0:
        step()
    th.cuda.synchronize()
    launches_per_step, _ = agent.q_net.ctx.read_timing()
    if dist is not None:
        dist.barrier()
        th.cuda.synchronize()
    # chain launches are event-timed on the library's stream.  An event record costs ~3.5 us of stream time (four records
    # around the two chain launches of a step: +14 us = 4 %), so short runs (the driver's --steps 20) bracket ONE launch per
    # step, the launches of a step taking turns, and long runs bracket all launches of every 4th step.
    timing_every = int(os.environ.get("MORL_BENCH_TIMING", -1 if steps <= 50 else 4))
    if not launches_per_step:
        timing_every = 1 if timing_every == -1 else timing_every
    agent.q_net.ctx.set_timing(timing_every)
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    t_enq = time.perf_counter() - t0                 # host time to enqueue the timed steps (no synchronisation inside)
    th.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        th.cuda.synchronize()
    wall = time.perf_counter() - t0
    n_chain, chain_ms = agent.q_net.ctx.read_timing()
    agent.q_net.ctx.set_timing(False)
    gpu_ms = e0.elapsed_time(e1)
    if dist is not None:
        t = th.tensor([wall], device=dev, dtype=th.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    res = {"wall": wall, "host_enqueue_ms_per_step": t_enq * 1e3 / steps, "gpu_ms_per_step_events": gpu_ms / steps,
           "n_chain": n_chain, "chain_ms": chain_ms, "timed_steps": (len(range(0, steps, timing_every)) if timing_every > 0 else steps if timing_every == -1 else 0),
           "launches_per_step": launches_per_step, "timing_mode": timing_every,
           "loss": agent.last_loss(), "engine": agent.q_net.ctx.engine, "W": W, "B": B}
    del agent
    return res


def _roofline(res, rows_rank):
    """Dominant kernel = mlp_chain (per step and rank: the forward passes -- one launch of three chains, or the slabs launch +
    the hoisted training forward of a sharded step -- and the backward-dX launch): algorithmic flop of the timed launches
    over their summed HIP-event duration."""
    n_chain, chain_ms, timed_steps = res["n_chain"], res["chain_ms"], res["timed_steps"]
    chain_flop_step = rows_rank * (3 * FWD_FLOP_ROW + BWD_DX_FLOP_ROW)
    launches_per_step = res.get("launches_per_step") or (n_chain / timed_steps if timed_steps else 0)
    flop_per_launch = chain_flop_step / launches_per_step if launches_per_step else float("nan")
    avg_launch_s = (chain_ms * 1e-3 / n_chain) if n_chain else float("nan")
    achieved = flop_per_launch / avg_launch_s / 1e12 if n_chain else float("nan")
    traffic = measured_chain_traffic()
    return {"bound": "mfma", "kernel": "mlp_chain (layer-fused Q-net forward / backward-dX)",
            "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/)",
            "traffic_source": "committed profile (profiles/*_pmc_summary.json: PMC counters cannot be read in-process); "
                              "constant of the repo, not of this run" if traffic is not None else None,
            "launches_timed": n_chain, "timed_steps": timed_steps, "launches_per_step": launches_per_step,
            "timing": ("one launch per step, taking turns" if res.get("timing_mode") == -1 else
                       "all launches of every %d-th step" % res.get("timing_mode", 0)),
            "avg_launch_us": avg_launch_s * 1e6,
            "algorithmic_flop_per_launch": flop_per_launch}


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
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="with --force-shard on one GPU: run the step of rank 0 of a job of this many ranks (its kernels, "
                         "launches, host work and message sizes; the other ranks' slabs are zeros) -- a measurement aid, "
                         "the line it prints is labelled as such")
    ap.add_argument("--shard-axis", choices=["auto", "batch", "weights"], default="auto",
                    help="how N > 1 ranks split one update: batch (B/N transitions per rank, one all-reduce), weights (W/N "
                         "weights per rank: all-gather of Q(w) + all-reduce, the north_star's description); auto = batch for "
                         "the strong-scaled job, weights for the weak-scaled one (its weight axis is what grows)")
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
    if not th.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    th.cuda.set_device(local_rank)
    dev = th.device("cuda", local_rank)
    dist = None
    if world > 1 or a.force_shard:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        print(f"[bench] rank {rank}/{world} on {dev}: RCCL nranks={dist.get_world_size()}", file=sys.stderr, flush=True)
    sharded = dist is not None

    if a.weights % world:
        raise SystemExit(f"--weights {a.weights} must be divisible by the number of ranks ({world})")
    B = a.batch
    W_head = a.weights * (world if (a.scaling == "weak" and world > 1) else 1)
    def axis_of(scaling_mode):
        return a.shard_axis if a.shard_axis != "auto" else ("weights" if scaling_mode == "weak" else "batch")
    head_axis = axis_of(a.scaling)
    if sharded and head_axis == "batch" and B % max(world, a.emulate_world if a.force_shard else 1):
        raise SystemExit(f"--batch {B} must be divisible by the number of ranks")
    head = run_job(a, dist, world, rank, dev, W_head, sharded, a.steps, a.warmup, head_axis)
    sub = None
    if world > 1 and not a.no_sub_record:
        # the other scaling mode, same steps / warm-up, reported as a sub-record of the same line
        W_sub = a.weights if a.scaling == "weak" else a.weights * world
        try:
            sub = run_job(a, dist, world, rank, dev, W_sub, sharded, a.steps, a.warmup,
                          axis_of("weak" if a.scaling == "strong" else "strong"))
        except Exception as exc:                      # the headline must survive a failing sub-record
            sub = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        def record(res, W, scaling):
            rows_step = B * W                             # TD rows per gradient step of the whole job
            ms = res["wall"] * 1e3 / a.steps
            return {"value": rows_step * a.steps / res["wall"], "unit": "TD-updates/s", "ms_per_step": ms,
                    "scaling": scaling, "weights": W, "weights_per_gpu": W // world,
                    "updates_per_s": a.steps / res["wall"],
                    "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"],
                    "gpu_ms_per_step_events": res["gpu_ms_per_step_events"], "last_loss": res["loss"],
                    "roofline": _roofline(res, rows_step // max(world, a.emulate_world if a.force_shard else 1)),
                    "whole_step_algorithmic_tflops": rows_step * (4 * FWD_FLOP_ROW + BWD_DX_FLOP_ROW) / (ms * 1e-3) / 1e12}

        scaling = "weak" if world == 1 else a.scaling     # (one GPU: per-GPU work is the metric's workload either way)
        h = record(head, W_head, scaling)
        out = {
            "metric": "Envelope-Q TD updates/sec (batch x weights x obj = 256 x 64 x 3)",
            "value": h["value"],
            "unit": "TD-updates/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": h["ms_per_step"],
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Envelope.update(): B={B} x W={W_head} x R={R}, obs {D}, {A} actions, net {ARCH}, "
                                   f"PER {'on' if a.per else 'off'}, buffer 20k seeded transitions (BASELINE.md s3)",
                       "global_batch": B, "weights": W_head, "objectives": R,
                       "weights_per_gpu": W_head // world,
                       "parallelism": "single GPU" if world == 1 else (
                           f"batch axis sharded over {world} GPUs, {B // world} transitions x {W_head} weights each "
                           f"({scaling} scaling; one RCCL all-reduce of gradient | loss | priorities)" if head_axis == "batch" else
                           f"weight axis sharded over {world} GPUs, {W_head // world} weights each ({scaling} scaling; RCCL "
                           "all-gather of Q(w), all-reduce of gradients)"),
                       "shard_axis": head_axis if (world > 1 or a.force_shard) else None,
                       "engine": head["engine"],
                       "setup": "0.5 s device clock ramp (dummy GEMMs) before the warm-up steps"},
            "updates_per_s": h["updates_per_s"],
            "scalar_td_per_s": h["value"] * R,
            "gpu_ms_per_step_events": h["gpu_ms_per_step_events"],
            "host_enqueue_ms_per_step": h["host_enqueue_ms_per_step"],
            "last_loss": h["last_loss"],
            "roofline": dict(h["roofline"], whole_step_algorithmic_tflops=h["whole_step_algorithmic_tflops"]),
        }
        if a.force_shard and a.emulate_world > 1 and world == 1:
            share = (f"{B // a.emulate_world} transitions x {W_head} weights" if head_axis == "batch"
                     else f"{B} transitions x {W_head // a.emulate_world} weights")
            out["emulated"] = (f"NOT a job throughput: the step of rank 0 of a {a.emulate_world}-rank job ({share}) run alone on "
                               "one GPU; value / ms_per_step describe that rank")
        if sub is not None:
            key = "weak_scaling" if scaling == "strong" else "strong_scaling"
            if "error" in sub:
                out[key] = sub
            else:
                other = "weak" if scaling == "strong" else "strong"
                out[key] = dict(record(sub, sub["W"], other),
                                note=f"sub-record, NOT the headline: {other} scaling, W = {sub['W']} sampled weights in total "
                                     f"({sub['W'] // world} per GPU; {axis_of(other)} axis sharded), same steps / warm-up, measured right after the headline")
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B, W_head, bool(a.per))
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), file=result_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
