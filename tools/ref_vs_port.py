"""BASELINE.md section 3 / VERDICT round 1 weak #1c: the CPU baseline bench.py reports is the oracle's PORT of the reference's
as-written algorithm (kind "port") because /root/reference cannot travel to the GPU box.  Where both exist (the build
container), this script times the UNMODIFIED reference ``Envelope.update()`` beside the port on the same synthetic workload
(B=256, W=64, obs 32, 3 objectives, net [256]*4, PER on) and prints the ratio.  CPU only; run from the repo root."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch as th

import ref_harness

B, W, D, A, R = 256, 64, 32, 6, 3
ARCH = [256, 256, 256, 256]


def time_reference(threads, n=3):
    ref = ref_harness.import_reference()
    th.set_num_threads(threads)
    th.manual_seed(0); np.random.seed(0)
    env = ref_harness.FakeEnv(obs_dim=D, n_actions=A, reward_dim=R)
    ag = ref.envelope.Envelope(env, learning_rate=3e-4, net_arch=ARCH, batch_size=B, gamma=0.99, max_grad_norm=1.0, tau=1.0,
                               target_net_update_freq=200, envelope=True, num_sample_w=W, per=True, per_alpha=0.6,
                               buffer_size=100_000, gradient_updates=1, log=False, seed=0, device="cpu")
    ref_harness.fill_buffer_synthetic(ag.replay_buffer, 2_000, D, A, R, seed=0)
    ag.global_step = 1001
    ag.update()                       # warm-up
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); ag.update(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def time_port(threads, n=3):
    import envelope_oracle as orc
    th.set_num_threads(threads)
    th.manual_seed(0)
    rng = np.random.default_rng(0)
    online = orc.init_qnet_params(D, A, R, ARCH)
    target = [p.clone() for p in online]
    m = [th.zeros_like(p) for p in online]
    v = [th.zeros_like(p) for p in online]
    mk = lambda: (th.tensor(rng.standard_normal((B, D)), dtype=th.float32), th.tensor(rng.integers(A, size=(B, 1)), dtype=th.uint8),
                  th.tensor(rng.standard_normal((B, R)), dtype=th.float32), th.tensor(rng.standard_normal((B, D)), dtype=th.float32),
                  th.tensor((rng.random((B, 1)) < 0.05), dtype=th.float32))
    ts = []
    for k in range(n + 1):
        sw = th.tensor(orc.random_weights(R, W, "gaussian", rng=rng), dtype=th.float32)
        t0 = time.perf_counter()
        orc.envelope_update(online, target, m, v, k + 1, mk(), sw, n_actions=A, reward_dim=R, dedup=False)
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts[1:]))


if __name__ == "__main__":
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(32, th.get_num_threads())
    r, p = time_reference(threads), time_port(threads)
    print(json.dumps({"threads": threads, "host_logical_cpus": os.cpu_count(), "reference_s_per_update": r, "port_s_per_update": p,
                      "reference_td_updates_per_s": B * W / r, "port_td_updates_per_s": B * W / p, "port_over_reference_time": p / r}))
