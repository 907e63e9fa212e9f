"""Per-step GPU time of the first Envelope.update() calls (development aid: why is a 5 + 20 step run slower than steady state?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as th
import bench
from morl_baselines_amd.envelope import Envelope
dev = th.device("cuda:0")
spin = float(os.environ.get("SPIN", "0"))
agent = Envelope(bench.SyntheticEnv(), learning_rate=3e-4, net_arch=bench.ARCH, batch_size=256, gamma=0.99, max_grad_norm=1.0, tau=1.0,
                 target_net_update_freq=200, envelope=True, num_sample_w=64, per=True, per_alpha=0.6, buffer_size=100_000,
                 gradient_updates=1, log=False, seed=0, device=dev)
bench.fill_buffer(agent.replay_buffer, 20_000, seed=0)
agent.global_step = 1001
if spin > 0:
    a = th.randn(4096, 4096, device=dev); t0 = time.time()
    while time.time() - t0 < spin:
        for _ in range(10): a @ a
        th.cuda.synchronize()
evs = [th.cuda.Event(enable_timing=True) for _ in range(61)]
t_host = []
evs[0].record()
for k in range(60):
    t0 = time.perf_counter(); agent.update(); agent.global_step += 1; t_host.append(time.perf_counter() - t0)
    evs[k + 1].record()
th.cuda.synchronize()
gpu = [evs[k].elapsed_time(evs[k + 1]) for k in range(60)]
print("spin", spin, "gpu ms per step:", " ".join("%.3f" % g for g in gpu))
print("host ms per step:", " ".join("%.3f" % (h * 1e3) for h in t_host))
