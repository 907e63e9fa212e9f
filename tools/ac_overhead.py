"""Where does a single-learner update spend its wall time?  (host enqueue vs GPU execution vs Python glue)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th
from morl_baselines_amd.ac_engine import ACEngine, ALGO_MOSAC
import morl_baselines_amd.ac_engine as ae

dev = th.device("cuda:0")
eng = ACEngine(ALGO_MOSAC, 11, 3, 3, [256, 256], action_low=-1.0, action_high=1.0, max_rows=128, device=dev, device_steps=True)
eng.q.normal_(0, 0.05); eng.pol.normal_(0, 0.05); eng.q_target.copy_(eng.q)
B = 128
obs, nobs = th.randn(1, B, 11, device=dev), th.randn(1, B, 11, device=dev)
act, rew, done = th.rand(1, B, 3, device=dev) * 2 - 1, th.randn(1, B, 3, device=dev), th.zeros(1, B, device=dev)
w = th.tensor([[0.3, 0.3, 0.4]], device=dev)
cfg = eng.make_cfg(q_lr=1e-3, policy_iters=2, autotune=True, target_entropy=-3.0)
eps = th.randn(5, 1, B, 3, device=dev)
c_time = [0.0]
orig = eng.lib.lib.morl_ac_update
def timed(*a):
    t0 = time.perf_counter(); r = orig(*a); c_time[0] += time.perf_counter() - t0; return r
class L:  # proxy
    def __getattr__(self, k): return timed if k == "morl_ac_update" else getattr(eng.lib.lib, k)
real = eng.lib.lib
for label in ("async", "sync each"):
    eng.lib.lib = L()
    c_time[0] = 0.0
    for _ in range(20):
        eng.update(cfg, obs=obs, actions=act, rewards=rew, next_obs=nobs, dones=done, w=w, eps_next=eps[0], eps_pi=eps[1:3], eps_alpha=eps[3:], want=())
    th.cuda.synchronize(); c_time[0] = 0.0
    t0 = time.perf_counter()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    N = 300
    for _ in range(N):
        eng.update(cfg, obs=obs, actions=act, rewards=rew, next_obs=nobs, dones=done, w=w, eps_next=eps[0], eps_pi=eps[1:3], eps_alpha=eps[3:], want=())
        if label != "async": th.cuda.synchronize()
    enq = time.perf_counter() - t0
    e1.record(); th.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"{label}: per update: wall {wall/N*1e6:.0f} us, host loop {enq/N*1e6:.0f} us, inside C call {c_time[0]/N*1e6:.0f} us, GPU events {e0.elapsed_time(e1)/N*1e3:.0f} us")
    eng.lib.lib = real
