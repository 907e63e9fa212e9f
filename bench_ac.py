"""Actor-critic update benchmark (the "next" rows of SURVEY.md section 8: C3 CAPQL, M1/M2 MOSAC / MORL-D, G7 GPI-PD
continuous) on one MI355X.  bench.py stays the north-star (Envelope) line the driver runs; this script measures the
widened rows with the same conventions:

    python bench_ac.py --workload capql|mosac|morld|gpipd|gpi|ens [--pop 64] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench_ac.py --workload morld --gpus N --pop 64      # the population's learners are independent units: pop / N per
                                                                # GPU, no data-path collective ("replicas only", weak scaling)

One "step" = one gradient update of every learner in the job (``update()`` body of the reference agent; for ``morld``
one pass of ``MORLD.__update_others`` over ``pop`` sub-problem learners, morld.py:423-433), on synthetic transitions
resident in HBM, reference-default shapes (batch 128, net [256, 256], twin critics).  ``value`` = learner-updates/s.
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

PEAK_FP32_MFMA_TFLOPS = 157.3
RESULT_OUT = sys.stdout

SHAPES = {  # obs dim, action dim, objectives of the environments BASELINE.json names
    "capql": dict(D=17, Ad=6, R=2, env="mo-halfcheetah-v4"),
    "mosac": dict(D=11, Ad=3, R=3, env="mo-hopper-v4"),
    "morld": dict(D=11, Ad=3, R=3, env="mo-hopper-v4"),
    "gpipd": dict(D=11, Ad=3, R=3, env="mo-hopper-v4"),
    "gpi": dict(D=7, Ad=6, R=3, env="mo-minecart-v0 (GPI-PD, discrete: 6 actions)"),
    "ens": dict(D=7, Ad=6, R=3, env="mo-minecart-v0 (GPI-PD Dyna model: one-hot action in, next-obs delta + reward out)"),
}
ARCH = [256, 256]
B = 128


def mlp_macs(dims):
    return sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def update_flop(workload, D, Ad, R, rows, policy_iters, do_policy=True):
    """Algorithmic flop of one learner's update: 2 * MACs per row for a forward, dX backward (all layers but the first,
    plus the first for the actor path) and dW backward; element-wise work not counted."""
    w_in = 0 if workload in ("mosac", "morld") else R
    heads = 1 if workload == "gpipd" else 2
    q = [D + Ad + w_in] + ARCH + [R]
    p = [D + w_in] + ARCH + [heads * Ad]
    fq, fp = mlp_macs(q), mlp_macs(p)
    dxq, dxp = fq - q[0] * q[1], fp - p[0] * p[1]
    macs = fp + 2 * fq                 # a' and the twin target critics
    macs += 2 * (fq + dxq + fq)        # twin critics: forward, dX, dW
    if do_policy:
        per_iter = fp + 2 * fq + 2 * fq + (dxp + fp)      # actor fwd, critics fwd, critics dX (incl. layer 0), actor bwd
        if workload in ("mosac", "morld"):
            per_iter += fp                                  # fresh log-prob for the alpha step
        macs += policy_iters * per_iter
    return 2.0 * macs * rows


def cpu_baseline(workload, shp, pop, budget_s=20.0):
    """The oracle (torch-CPU restatement of the reference update) on this box's host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ac_oracle as ac

    D, Ad, R = shp["D"], shp["Ad"], shp["R"]
    gen = th.Generator().manual_seed(0)
    rnd = lambda *s: th.randn(*s, generator=gen)  # noqa: E731
    w_in = workload not in ("mosac", "morld")
    qspec = ac.MlpSpec(D + Ad + (R if w_in else 0), tuple(ARCH), R, layer_norm=(workload == "gpipd"),
                       drop_rate=0.01 if workload == "gpipd" else 0.0)
    trunk = ac.MlpSpec(D + (R if w_in else 0), tuple(ARCH))
    heads = 1 if workload == "gpipd" else 2
    q = [ac.init_mlp_params(qspec, gen) for _ in range(2)]
    tq = [ac.clone(n) for n in q]
    pol = ac.init_mlp_params(trunk, gen)
    for _ in range(heads):
        pol += [rnd(Ad, ARCH[-1]) * 0.05, th.zeros(Ad)]
    tpol = ac.clone(pol)
    qs = dict(exp_avg=ac.zeros_like(q[0] + q[1]), exp_avg_sq=ac.zeros_like(q[0] + q[1]))
    ps = dict(exp_avg=ac.zeros_like(pol), exp_avg_sq=ac.zeros_like(pol))
    als = dict(exp_avg=[th.zeros(1)], exp_avg_sq=[th.zeros(1)])
    la = th.zeros(1)
    scale, bias = th.ones(Ad), th.zeros(Ad)
    wv = th.softmax(rnd(B, R), dim=1)
    obs, act, rew, nobs, done = rnd(B, D), th.tanh(rnd(B, Ad)), rnd(B, R), rnd(B, D), (th.rand(B, 1, generator=gen) < 0.05).float()

    def one(step):
        if workload == "capql":
            ac.capql_update(qspec, trunk, q, tq, pol, qs, ps, (obs, act, wv, rew, nobs, done.reshape(-1)), rnd(B, Ad),
                            rnd(B, Ad), scale, bias, gamma=0.99, alpha=0.2, lr=3e-4, tau=0.005, step=step)
        elif workload in ("mosac", "morld"):
            ac.mosac_update(qspec, trunk, q, tq, pol, la, qs, ps, als, (obs, act, rew, nobs, done), wv[0], rnd(B, Ad),
                            [rnd(B, Ad), rnd(B, Ad)], [rnd(B, Ad), rnd(B, Ad)], scale, bias, gamma=0.99, tau=0.005,
                            q_lr=1e-3, policy_lr=3e-4, q_step=step, a_step=2 * step - 1, policy_freq=2, do_policy=True,
                            do_target=True, autotune=True, alpha=float(la.exp()), target_entropy=-float(Ad))
        else:
            rows = 2 * B
            rep = lambda x: x.repeat(2, 1)  # noqa: E731
            keep = lambda: [(th.rand(rows, h, generator=gen) >= 0.01).float() for h in ARCH]  # noqa: E731
            drop = {k: [keep(), keep()] for k in ("target", "q", "q_pi")}
            ac.gpipd_cont_update(qspec, trunk, q, tq, pol, tpol, qs, ps, [rep(obs), rep(act), rep(rew), rep(nobs), rep(done)],
                                 rep(wv), rnd(rows, Ad), drop, scale, bias, gamma=0.99, lr=3e-4, tau=0.005, q_step=step,
                                 p_step=step, do_policy=True, n_per=B)

    # these nets are small: torch's default (all cores) is far from the best thread count -- probe a few and keep the best
    one(1)
    step, best = 2, (None, float("inf"))
    for nt in (1, 4, 16, th.get_num_threads()):
        th.set_num_threads(nt)
        one(step); step += 1
        t0 = time.perf_counter()
        for _ in range(3):
            one(step); step += 1
        dt = (time.perf_counter() - t0) / 3
        if dt < best[1]:
            best = (nt, dt)
    th.set_num_threads(best[0])
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 300:
        t0 = time.perf_counter()
        one(step)
        times.append(time.perf_counter() - t0)
        step += 1
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "learner-updates/s", "cores": best[0], "kind": "port",
            "sample": f"{len(times)} timed single-learner updates of oracle/ac_oracle.py on torch-CPU at the best of 1/4/16/all "
                      f"threads ({best[0]}), median; the reference advances a population of {pop} sequentially, i.e. at this rate"}


def bench_gpi(a):
    """GPI-PD with discrete actions: one ``morl_gpi_update`` (batch 128 doubled to 256 rows, ensemble of 2 conditioned
    Q-nets [256]*4 with LayerNorm + Dropout, envelope target over K = 5 weights) per step."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from morl_baselines_amd.gpi_engine import GPIEngine

    dev = th.device("cuda", 0)
    shp = SHAPES["gpi"]
    D, A, R, K, rows, arch = shp["D"], shp["Ad"], shp["R"], 5, 2 * B, [256, 256, 256, 256]
    eng = GPIEngine(D, A, R, arch, max_rows=rows, max_support=K, device=dev)
    gen = th.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: th.randn(*s, generator=gen, device=dev)  # noqa: E731
    with th.no_grad():
        eng.q.copy_(rnd(*eng.q.shape) * 0.05)
        for n in range(2):
            v = eng.views(eng.q, n)
            for k in (6, 10, 14):
                v[k].fill_(1.0)
        eng.q_target.copy_(eng.q)
    obs, nobs, rew = rnd(rows, D), rnd(rows, D), rnd(rows, R)
    act = th.randint(0, A, (rows,), generator=gen, device=dev)
    done = (th.rand(rows, generator=gen, device=dev) < 0.05).float()
    w = th.softmax(rnd(rows, R), dim=-1).contiguous()
    sw = th.softmax(rnd(K, R), dim=-1).contiguous()
    st = {"n": 0}

    def step():
        st["n"] += 1
        eng.update(obs=obs, actions=act, rewards=rew, next_obs=nobs, dones=done, w=w, sampled_w=sw, adam_step=st["n"],
                   gpi_pd=True, n_per=B, dropout_seed=st["n"], want=("critic_loss", "gtd_error"))

    for _ in range(a.warmup):
        step()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    th.cuda.synchronize()
    wall = time.perf_counter() - t0
    macs = D * 256 + 3 * 256 * 256 + 256 * A * R             # per row and net (the R-input weight embedding not counted)
    flop = 2.0 * macs * 2 * (rows + rows * K + 3 * rows)      # target, envelope target, online forward + dX + dW
    ms = wall * 1e3 / a.steps
    tf = flop / (ms * 1e-3) / 1e12
    out = {"metric": "GPI-PD gradient updates/sec", "value": a.steps / wall, "unit": "updates/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"GPIPD.update() body: batch {B} doubled to {rows} rows, 2 conditioned Q-nets {arch} "
                                  f"(LayerNorm, Dropout 0.01), envelope target over {K} weights, PER errors; shapes of "
                                  f"{shp['env']}"},
           "roofline": {"bound": "mfma", "kernel": "gemm_batched / gemm_wave_batched", "achieved": tf,
                        "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                        "note": "algorithmic GEMM flop of the update / wall time (launch-latency-bound workload)"},
           "algorithmic_flop_per_step": flop}
    # the reference's default configuration: prioritised replay, 20 gradient updates per environment step -- ONE library entry
    # (morl_gpi_update_n_per: per iteration tree descent + gather, update, max(|gtd|, min) ** alpha back into the tree)
    from morl_baselines_amd.replay import PrioritizedReplayBuffer
    G = 20
    buf = PrioritizedReplayBuffer((D,), 1, rew_dim=R, max_size=32768, action_dtype=np.int32, device=dev)
    rng = np.random.default_rng(0)
    buf.add_batch(rng.standard_normal((20000, D)).astype(np.float32), rng.integers(0, A, (20000, 1)).astype(np.int32),
                  rng.standard_normal((20000, R)).astype(np.float32), rng.standard_normal((20000, D)).astype(np.float32),
                  (rng.random((20000, 1)) < 0.05).astype(np.float32))
    sc = (th.empty(rows, D, device=dev), th.empty(rows, dtype=th.int32, device=dev), th.empty(rows, R, device=dev),
          th.empty(rows, D, device=dev), th.empty(rows, device=dev))

    def per_step():
        items = []
        for _ in range(G):
            st["n"] += 1
            items.append(dict(obs=sc[0], actions=sc[1], rewards=sc[2], next_obs=sc[3], dones=sc[4], w=w, sampled_w=sw,
                              adam_step=st["n"], gpi_pd=True, n_per=B, dropout_seed=st["n"],
                              want=("critic_loss", "td_error", "gtd_error")))
        eng.update_n_per(items, buffer=buf, u01=np.random.random_sample((G, B)), doubled=True, use_gtd=True, alpha=0.6,
                         min_priority=0.01)

    for _ in range(3):
        per_step()
    th.cuda.synchronize()
    n_loops = max(3, a.steps // G)
    enq, t0 = [], time.perf_counter()
    for _ in range(n_loops):
        t1 = time.perf_counter()
        per_step()
        enq.append(time.perf_counter() - t1)
    th.cuda.synchronize()
    loop_ms = (time.perf_counter() - t0) * 1e3 / n_loops
    out["per_loop"] = {"what": f"GPIPD.update() with per=True: {G} gradient updates per environment step through ONE library "
                               "entry (morl_gpi_update_n_per: per iteration sum-tree descent + gather, update, priorities "
                               "back into the tree); buffer of 20 000 transitions",
                       "ms_per_env_step": loop_ms, "ms_per_update": loop_ms / G,
                       "host_enqueue_ms_per_env_step": float(np.median(enq)) * 1e3, "library_entries_per_env_step": 1}
    if not a.no_cpu_baseline:
        import gpi_oracle as go
        from ac_oracle import clone

        spec = go.GpiSpec(D, R, A, tuple(arch), True, 0.01)
        q = [[p.cpu().clone() for p in eng.views(eng.q, n)] for n in range(2)]
        tq = [clone(n) for n in q]
        flat = [p for n in q for p in n]
        state = dict(exp_avg=[th.zeros_like(p) for p in flat], exp_avg_sq=[th.zeros_like(p) for p in flat])
        batch = [obs.cpu(), act.cpu().float().reshape(-1, 1), rew.cpu(), nobs.cpu(), done.cpu().reshape(-1, 1)]
        g2 = th.Generator().manual_seed(0)
        keep = lambda n_: [[(th.rand(n_, h, generator=g2) >= 0.01).float() for h in arch[1:]] for _ in range(2)]  # noqa: E731
        k, best = 0, (None, float("inf"))
        for nt in (1, 4, 16, th.get_num_threads()):        # small nets: all cores is not the fastest setting
            th.set_num_threads(nt)
            drop = dict(target=keep(rows), env=keep(rows * K), q=keep(rows))
            t1 = time.perf_counter()
            for _ in range(2):
                k += 1
                go.gpi_update(spec, q, tq, state, batch, w.cpu(), sw.cpu(), drop, gamma=0.99, lr=3e-4, step=k,
                              min_priority=0.01, gpi_pd=True, n_per=B)
            dt = (time.perf_counter() - t1) / 2
            if dt < best[1]:
                best = (nt, dt)
        th.set_num_threads(best[0])
        times = []
        t_end = time.perf_counter() + 15.0
        while time.perf_counter() < t_end and len(times) < 100:
            k += 1
            drop = dict(target=keep(rows), env=keep(rows * K), q=keep(rows))
            t1 = time.perf_counter()
            go.gpi_update(spec, q, tq, state, batch, w.cpu(), sw.cpu(), drop, gamma=0.99, lr=3e-4, step=k, min_priority=0.01,
                          gpi_pd=True, n_per=B)
            times.append(time.perf_counter() - t1)
        med = float(np.median(times[1:] or times))
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "updates/s", "cores": best[0], "kind": "port",
                               "sample": f"{len(times)} timed updates of oracle/gpi_oracle.py on torch-CPU at the best of "
                                         f"1/4/16/all threads ({best[0]}), median"}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), file=RESULT_OUT, flush=True)


def bench_ens(a):
    """Dyna model of GPI-PD: one optimiser step of ``ProbabilisticEnsemble.fit`` (probabilistic_ensemble.py:248-252) --
    5 members x batch 256, net [256, 256, 256] (gpi_pd.py defaults), Gaussian NLL with bounded log-variance, Adam with
    per-layer weight decay -- per step."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from morl_baselines_amd.dynamics import DECAYS, ProbabilisticEnsemble

    dev = th.device("cuda", 0)
    shp = SHAPES["ens"]
    E, rows, arch = 5, 256, [256, 256, 256]
    din, dout = shp["D"] + shp["Ad"], shp["D"] + shp["R"]
    th.manual_seed(0)
    ens = ProbabilisticEnsemble(din, dout, ensemble_size=E, arch=arch, device=dev, max_rows=rows)
    ens.decays = list(DECAYS)
    gen = th.Generator(device=dev).manual_seed(0)
    x = th.randn(E, rows, din, generator=gen, device=dev)
    y = th.randn(E, rows, dout, generator=gen, device=dev) * 0.1
    ens.inputs_mu = th.zeros((1, din), device=dev)
    ens.inputs_sigma = th.ones((1, din), device=dev)
    p0 = ens.flat.clone()
    for _ in range(a.warmup):
        ens.train_step(x, y)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ens.train_step(x, y)
    th.cuda.synchronize()
    wall = time.perf_counter() - t0
    dims = [din] + arch + [2 * dout]
    macs = sum(i * o for i, o in zip(dims[:-1], dims[1:]))
    flop = 2.0 * macs * E * rows * 3 - 2.0 * dims[0] * dims[1] * E * rows     # forward + dX + dW (no dX for the input layer)
    ms = wall * 1e3 / a.steps
    tf = flop / (ms * 1e-3) / 1e12
    out = {"metric": "dynamics-ensemble optimiser steps/sec", "value": a.steps / wall, "unit": "steps/s", "n_gpus": 1,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"ProbabilisticEnsemble.fit() inner step: {E} members x batch {rows}, net {arch}, "
                                  f"in {din} / out 2x{dout}; shapes of {shp['env']}"},
           "roofline": {"bound": "mfma", "kernel": "gemm_batched / gemm_wave_batched", "achieved": tf,
                        "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                        "note": "algorithmic GEMM flop of the step / wall time (launch-latency-bound workload)"},
           "algorithmic_flop_per_step": flop}
    if not a.no_cpu_baseline:
        import ens_oracle as eo

        views = ens._layer_views(p0)
        state = dict(W=[w.transpose(1, 2).contiguous().cpu() for w, _ in views],
                     b=[b.reshape(E, 1, -1).contiguous().cpu() for _, b in views],
                     max_lv=th.ones(1, dout) / 2.0, min_lv=-th.ones(1, dout) * 10.0, mu=th.zeros(1, din),
                     sigma=th.ones(1, din))
        order = [p for pair in zip(state["W"], state["b"]) for p in pair] + [state["max_lv"], state["min_lv"]]
        state["m"], state["v"] = [th.zeros_like(p) for p in order], [th.zeros_like(p) for p in order]
        xc, yc = x.cpu(), y.cpu()
        k, best = 0, (None, float("inf"))
        for nt in (1, 4, 16, th.get_num_threads()):
            th.set_num_threads(nt)
            t1 = time.perf_counter()
            for _ in range(3):
                k += 1
                eo.train_step(state, xc, yc, k)
            dt = (time.perf_counter() - t1) / 3
            if dt < best[1]:
                best = (nt, dt)
        th.set_num_threads(best[0])
        times, t_end = [], time.perf_counter() + 15.0
        while time.perf_counter() < t_end and len(times) < 200:
            k += 1
            t1 = time.perf_counter()
            eo.train_step(state, xc, yc, k)
            times.append(time.perf_counter() - t1)
        med = float(np.median(times[1:] or times))
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "steps/s", "cores": best[0], "kind": "port",
                               "sample": f"{len(times)} timed steps of oracle/ens_oracle.py::train_step on torch-CPU at the "
                                         f"best of 1/4/16/all threads ({best[0]}), median"}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), file=RESULT_OUT, flush=True)


def _claim_stdout():
    """stdout must carry exactly ONE line, rank 0's JSON: C libraries (RCCL prints a version banner to stdout, flushed
    at exit, i.e. after the JSON) and the other ranks are moved to stderr; the result is written to the saved descriptor."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def bench_morld_contexts(a, D, Ad, R, pop, rows, shp):
    """``MORLD(devices=[...])``'s hot path (``__update_others``, morld.py:423-433) with the population split over ``a.devices`` contexts
    of ONE process: context g holds pop / G learners in its own ACEngine on GPU min(g, visible - 1); a step enqueues every context's
    update (independent learners: no exchange), the step ends when all devices are done."""
    from morl_baselines_amd.ac_engine import ALGO_MOSAC, ACEngine
    G = a.devices
    if pop % G:
        raise SystemExit(f"--pop {pop} must be divisible by --devices {G}")
    visible = th.cuda.device_count()
    devs = [th.device("cuda", min(g, visible - 1)) for g in range(G)]
    iters, ctxs = 2, []
    for g, dev in enumerate(devs):
        n = pop // G
        with th.cuda.device(dev):
            eng = ACEngine(ALGO_MOSAC, D, Ad, R, ARCH, action_low=-1.0, action_high=1.0, max_rows=rows, population=n, device=dev,
                           device_steps=True)
            gen = th.Generator(device=dev).manual_seed(g)
            rnd = lambda *s_, gen=gen, dev=dev: th.randn(*s_, generator=gen, device=dev)  # noqa: E731
            with th.no_grad():
                eng.q.copy_(rnd(*eng.q.shape) * 0.05)
                eng.pol.copy_(rnd(*eng.pol.shape) * 0.05)
                eng.q_target.copy_(eng.q)
            data = dict(obs=rnd(n, rows, D), next_obs=rnd(n, rows, D), actions=th.tanh(rnd(n, rows, Ad)), rewards=rnd(n, rows, R),
                        dones=(th.rand(n, rows, generator=gen, device=dev) < 0.05).float(),
                        w=th.softmax(rnd(n, 1, R), dim=-1).contiguous())
        ctxs.append((dev, eng, data, n))

    def step():
        for dev, eng, data, n in ctxs:
            with th.cuda.device(dev):
                cfg = eng.make_cfg(q_lr=1e-3, policy_iters=iters, autotune=True, target_entropy=-float(Ad))
                eps = th.randn((1 + 2 * iters, n, rows, Ad), dtype=th.float32, device=dev)
                eng.update(cfg, eps_next=eps[0], eps_pi=eps[1:1 + iters], eps_alpha=eps[1 + iters:], want=("critic_loss",), **data)

    sync = lambda: [th.cuda.synchronize(d) for d in set(devs)]  # noqa: E731
    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    wall = time.perf_counter() - t0
    ms = wall * 1e3 / a.steps
    flop = update_flop("morld", D, Ad, R, rows, iters) * pop
    tf = flop / (ms * 1e-3) / 1e12
    distinct = len(set(devs))
    out = {"metric": "actor-critic learner updates/sec", "value": pop * a.steps / wall, "unit": "learner-updates/s",
           "rows_per_s": pop * rows * a.steps / wall, "n_gpus": distinct, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"morld: {pop} learners x batch {B} ({rows} rows), net {ARCH}, shapes of {shp['env']}; one pass of "
                                  f"MORLD.__update_others per step, the population split over {G} device contexts of one process",
                      "population": pop, "contexts": G, "learners_per_context": pop // G, "distinct_gpus": distinct,
                      "parallelism": f"{G} contexts (one ACEngine each) on {distinct} GPU(s); independent learners, no data-path collective",
                      "measured": (f"all {G} contexts on their own GPU" if distinct == G else
                                   f"FUNCTIONAL beyond {distinct} GPU(s): this box shows {visible}, so {G - distinct + 1} contexts share "
                                   "a device -- the multi-GPU figure is UNMEASURED")},
           "roofline": {"bound": "mfma", "kernel": "gemm_batched (exact-fp32 MFMA layers of all nets / learners)", "achieved": tf,
                        "peak": PEAK_FP32_MFMA_TFLOPS * distinct, "unit": "TFLOP/s", "frac": tf / (PEAK_FP32_MFMA_TFLOPS * distinct),
                        "traffic": None},
           "algorithmic_flop_per_step": flop}
    print(json.dumps(out), file=RESULT_OUT, flush=True)


def main():
    global RESULT_OUT
    RESULT_OUT = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="morld", choices=sorted(SHAPES))
    ap.add_argument("--pop", type=int, default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-mode", type=int, default=0, help="0 auto, 1 LDS tiles, 2 wave tiles (morl_ac_set_gemm_mode)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--devices", type=int, default=1,
                    help="morld in ONE process: the population split over this many device contexts (MORLD(devices=[...]): pop / N "
                         "learners per ACEngine, one context per visible GPU -- all on cuda:0 when the box has fewer, and the record "
                         "says how many distinct GPUs it really used)")
    ap.add_argument("--force-dp", action="store_true",
                    help="capql on ONE rank through the data-parallel path (gradient hook + RCCL all-reduce with itself)")
    a = ap.parse_args()
    if not th.cuda.is_available():
        raise SystemExit("bench_ac.py needs an MI355X (no CPU fallback exists)")
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus}")
    dev = th.device("cuda", local_rank)
    th.cuda.set_device(dev)
    dist = None
    if world > 1 or a.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if world > 1 and a.workload not in ("morld", "capql"):
            raise SystemExit("--gpus N: the MORL/D population shards as independent learners (no collective); CAPQL runs "
                             "data-parallel (gradient all-reduce inside the update)")
    if a.workload == "gpi":
        return bench_gpi(a)
    if a.workload == "ens":
        return bench_ens(a)
    from morl_baselines_amd.ac_engine import ALGO_CAPQL, ALGO_MOSAC, ALGO_TD3, ACEngine

    wl, shp = a.workload, SHAPES[a.workload]
    D, Ad, R = shp["D"], shp["Ad"], shp["R"]
    pop = a.pop if a.pop is not None else (64 if wl == "morld" else 1)
    pop_total = pop
    dp = (world > 1 or a.force_dp) and wl == "capql"         # data-parallel: every rank B rows of the job's world * B batch, grads averaged
    if world > 1 and not dp:
        if pop % world:
            raise SystemExit(f"--pop {pop} must be divisible by the number of GPUs")
        pop = pop // world                       # this rank's learners
    algo = {"capql": ALGO_CAPQL, "mosac": ALGO_MOSAC, "morld": ALGO_MOSAC, "gpipd": ALGO_TD3}[wl]
    rows = 2 * B if wl == "gpipd" else B
    if a.devices > 1:
        if wl != "morld" or world > 1:
            raise SystemExit("--devices N is the single-process MORL/D population over N device contexts")
        return bench_morld_contexts(a, D, Ad, R, pop, rows, shp)
    eng = ACEngine(algo, D, Ad, R, ARCH, action_low=-1.0, action_high=1.0, max_rows=rows, population=pop, device=dev,
                   q_layer_norm=(wl == "gpipd"), q_drop_rate=(0.01 if wl == "gpipd" else 0.0), device_steps=True)
    eng.lib.check(eng.lib.lib.morl_ac_set_gemm_mode(a.gemm_mode))
    gen = th.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: th.randn(*s, generator=gen, device=dev)  # noqa: E731
    with th.no_grad():                          # orthogonal-ish scale; the values do not matter for timing
        eng.q.copy_(rnd(*eng.q.shape) * 0.05)
        eng.pol.copy_(rnd(*eng.pol.shape) * 0.05)
        if eng.layer_norm:
            for p_ in range(pop):
                for n in range(2):
                    v = eng.q_views(eng.q, p_, n)
                    for k in (2, 6):
                        v[k].fill_(1.0)
        eng.q_target.copy_(eng.q)
        if eng.pol_target is not None:
            eng.pol_target.copy_(eng.pol)
    obs, nobs = rnd(pop, rows, D), rnd(pop, rows, D)
    act, rew = th.tanh(rnd(pop, rows, Ad)), rnd(pop, rows, R)
    done = (th.rand(pop, rows, generator=gen, device=dev) < 0.05).float()
    w = th.softmax(rnd(pop, rows if eng.w_input else 1, R), dim=-1).contiguous()
    iters = 2 if algo == ALGO_MOSAC else 1
    state = {"seed": 0}
    sync = None
    if dp:
        from morl_baselines_amd.distributed import average_gradients
        sync = average_gradients(dist)
        for buf in (eng.q, eng.pol, eng.q_target):
            dist.broadcast(buf, src=0)

    def step():
        state["seed"] += 1
        cfg = eng.make_cfg(q_lr=1e-3 if algo == ALGO_MOSAC else 3e-4, policy_iters=iters, autotune=(algo == ALGO_MOSAC),
                           target_entropy=-float(Ad), n_per=(B if wl == "gpipd" else 0), dropout_seed=state["seed"])
        eps = th.randn((1 + 2 * iters, pop, rows, Ad), dtype=th.float32, device=dev)      # the host agents draw these too
        eng.update(cfg, obs=obs, actions=act, rewards=rew, next_obs=nobs, dones=done, w=w, eps_next=eps[0],
                   eps_pi=eps[1:1 + iters], eps_alpha=eps[1 + iters:], want=("critic_loss",), grad_sync=sync)

    for _ in range(a.warmup):
        step()
    th.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    th.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = th.tensor([wall], device=dev, dtype=th.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        pop = pop_total
        if dp:
            rows = rows * world              # the job's batch
        if rank != 0:
            dist.destroy_process_group()
            return
    ms = wall * 1e3 / a.steps
    flop = update_flop(wl, D, Ad, R, rows, iters) * pop
    tf = flop / (ms * 1e-3) / 1e12
    out = {
        "metric": "actor-critic learner updates/sec", "value": pop * a.steps / wall, "unit": "learner-updates/s",
        "rows_per_s": pop * rows * a.steps / wall,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak" if (world == 1 or dp) else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{wl}: {pop} learner(s) x batch {B} ({rows} rows), net {ARCH}, twin critics, shapes of "
                               f"{shp['env']} (obs {D}, act {Ad}, {R} objectives); one morl_ac_update per step",
                   "population": pop, "rows": rows,
                   "parallelism": "single GPU" if world == 1 else
                                  (f"data-parallel over {world} GPUs: {rows // world} rows each, critic and actor gradients "
                                   "all-reduced (RCCL) inside the update, identical Adam steps" if dp else
                                   f"population split over {world} GPUs ({pop // world} learners each), independent "
                                   "replicas, no data-path collective")},
        "roofline": {"bound": "mfma", "kernel": "gemm_batched (exact-fp32 MFMA layers of all nets / learners)",
                     "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS,
                     "traffic": None,
                     "note": "algorithmic GEMM flop of the whole update / wall time of the step (launch-latency included): "
                             "a lower bound of the kernels' own rate; per-kernel durations in profiles/"},
        "algorithmic_flop_per_step": flop,
    }
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(wl, shp, pop)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out), file=RESULT_OUT, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
