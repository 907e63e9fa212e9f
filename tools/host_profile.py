"""Host-side cost of one Envelope step (Python + ctypes + launch enqueue), with cProfile: the single-GPU step or the step of one
rank of a sharded job (--emulate-world N on one GPU).  Run on the GPU box: python tools/host_profile.py [--weights W]
[--emulate-world N] [--steps K]."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", type=int, default=64)
    ap.add_argument("--emulate-world", type=int, default=0)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--top", type=int, default=35)
    a = ap.parse_args()
    from morl_baselines_amd.envelope import Envelope
    dev = th.device("cuda", 0)
    th.manual_seed(0)
    np.random.seed(0)
    agent = Envelope(bench.SyntheticEnv(), learning_rate=3e-4, net_arch=bench.ARCH, batch_size=256, gamma=0.99, max_grad_norm=1.0,
                     tau=1.0, target_net_update_freq=200, envelope=True, num_sample_w=a.weights, per=True, per_alpha=0.6,
                     buffer_size=100_000, gradient_updates=1, log=False, seed=0, device=dev)
    bench.fill_buffer(agent.replay_buffer, 20_000, seed=0)
    if a.emulate_world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from morl_baselines_amd.distributed import shard_envelope_agent
        shard_envelope_agent(agent, dist, emulate=(a.emulate_world, 0))
    agent.global_step = 1001

    def step():
        agent.update()
        agent.global_step += 1

    for _ in range(30):
        step()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enq = time.perf_counter() - t0
    th.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"weights {a.weights} emulate {a.emulate_world}: host enqueue {t_enq / a.steps * 1e3:.4f} ms/step, wall {t_all / a.steps * 1e3:.4f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    pr.disable()
    th.cuda.synchronize()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(a.top)
    print(out.getvalue())


if __name__ == "__main__":
    main()
