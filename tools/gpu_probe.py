"""GPU-box probe: flagship-size parity vs the oracle + raw timings of the update and of the forward pass.

    python tools/gpu_probe.py [--steps N] [--no-check] [--W 64] [--B 256]
Writes a JSON summary to gpurun_out/probe.json (and prints it).  Development tool, not part of the product.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

import numpy as np
import torch as th

import morl_baselines_amd.ops as ops
from morl_baselines_amd.native import load_library


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--W", type=int, default=64)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "probe.json"))
    a = ap.parse_args()
    D, A, R, arch = 32, 6, 3, (256, 256, 256, 256)
    B, W = a.B, a.W
    dev = th.device("cuda:0")
    lib = load_library()
    rng = np.random.default_rng(0)
    import envelope_oracle as orc
    g = th.Generator().manual_seed(0)
    online = orc.init_qnet_params(D, A, R, arch, generator=g)
    online = [p + 0.02 * th.randn(p.shape, generator=g) for p in online]
    target = [p + 0.02 * th.randn(p.shape, generator=g) for p in online]
    flat = lambda ps: th.cat([p.reshape(-1) for p in ps])
    obs = th.tensor(rng.standard_normal((B, D)), dtype=th.float32)
    nobs = th.tensor(rng.standard_normal((B, D)), dtype=th.float32)
    act = th.tensor(rng.integers(A, size=(B, 1)), dtype=th.uint8)
    rew = th.tensor(rng.standard_normal((B, R)), dtype=th.float32)
    done = th.tensor((rng.random((B, 1)) < 0.05), dtype=th.float32)
    sw = th.tensor(orc.random_weights(R, W, "gaussian", rng=rng), dtype=th.float32)
    ctx = ops.QNetContext(D, R, A, arch, B, W, lib=lib, fused=a.fused)
    P = ctx.n_params
    po, pt = flat(online).to(dev), flat(target).to(dev)
    m, v, gr = th.zeros(P, device=dev), th.zeros(P, device=dev), th.zeros(P, device=dev)
    d_obs, d_nobs, d_rew = obs.to(dev), nobs.to(dev), rew.to(dev)
    d_act, d_done, d_sw = act.reshape(-1).int().to(dev), done.reshape(-1).to(dev), sw.to(dev)
    kw = dict(gamma=0.99, lr=3e-4, max_grad_norm=1.0)
    out = {"B": B, "W": W, "device": th.cuda.get_device_name(0), "n_params": P, "engine": ctx.engine}

    if not a.no_check:
        res = ops.envelope_update(ctx, po, pt, gr, m, v, d_obs, d_nobs, d_act, d_rew, d_done, d_sw, adam_step=1, debug=True, **kw)
        th.cuda.synchronize()
        th.set_num_threads(os.cpu_count() or 1)
        t0 = time.time()
        mo = [th.zeros_like(p) for p in online]; vo = [th.zeros_like(p) for p in online]
        o = orc.envelope_update(online, target, mo, vo, 1, (obs, act, rew, nobs, done), sw, n_actions=A, reward_dim=R,
                                dedup=True, **kw)
        out["oracle_dedup_s"] = time.time() - t0
        rel = lambda x, y: float((x.cpu().double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30))
        out["qo_bit_identical"] = bool(th.equal(res["q_online_next"].cpu(), o["qo"]))
        out["qo_rel"] = rel(res["q_online_next"], o["qo"])
        out["qt_rel"] = rel(res["q_target_next"], o["qt"])
        out["qv_rel"] = rel(res["q_values"], o["q_values"])
        out["qv_bit_identical"] = bool(th.equal(res["q_values"].cpu(), o["q_values"]))
        out["pref_mismatch"] = int((res["pref"].cpu().long() != o["pref"]).sum())
        out["ac_mismatch"] = int((res["ac"].cpu().long() != o["ac"]).sum())
        out["loss"] = [res["loss"].item(), o["loss"].item()]
        out["grad_norm"] = [res["grad_norm"].item(), o["grad_norm"].item()]
        out["grads_rel"] = rel(gr, flat(o["grads"]))
        out["param_maxabs_diff"] = float((po.cpu() - flat(online)).abs().max())
        out["prio_rel"] = rel(res["priority"], o["priority_raw"])

    step = [2]
    def one():
        ops.envelope_update(ctx, po, pt, gr, m, v, d_obs, d_nobs, d_act, d_rew, d_done, d_sw, adam_step=step[0], **kw)
        step[0] += 1
    for _ in range(a.warmup):
        one()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        one()
    e1.record()
    th.cuda.synchronize()
    wall = time.perf_counter() - t0
    out["update_ms_events"] = e0.elapsed_time(e1) / a.steps
    out["update_ms_wall"] = wall * 1e3 / a.steps
    out["updates_per_s"] = a.steps / wall
    flop = 5 * B * W * 420352
    out["tflops_algorithmic"] = flop / (out["update_ms_events"] * 1e-3) / 1e12
    # forward only (build_input + 5 GEMMs)
    for _ in range(5):
        ops.qnet_forward(ctx, po, d_nobs, d_sw, 0)
    th.cuda.synchronize()
    e0.record()
    for _ in range(50):
        ops.qnet_forward(ctx, po, d_nobs, d_sw, 0)
    e1.record()
    th.cuda.synchronize()
    out["forward_ms"] = e0.elapsed_time(e1) / 50
    out["forward_tflops"] = B * W * 420352 / (out["forward_ms"] * 1e-3) / 1e12
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
