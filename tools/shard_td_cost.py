"""Development timing (not a bench line): what does one rank's stage B cost when the envelope arg-max runs over the
gathered candidates of G ranks (weak scaling: W_total = 64 * G) instead of its own 64?  One MI355X, synthetic slabs."""
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morl_baselines_amd import ops  # noqa: E402
from morl_baselines_amd.native import load_library  # noqa: E402

dev = th.device("cuda:0")
lib = load_library()
B, Wl, D, A, R = 256, 64, 32, 6, 3
ctx = ops.QNetContext(D, R, A, (256, 256, 256, 256), B, Wl, lib=lib)
P = ctx.n_params
g = th.Generator(device=dev).manual_seed(0)
po = th.randn(P, generator=g, device=dev) * 0.05
grads = th.zeros(P, device=dev)
obs = th.randn(B, D, generator=g, device=dev)
act = th.randint(0, A, (B,), generator=g, device=dev).to(th.int32)
rew, done = th.randn(B, R, generator=g, device=dev), th.zeros(B, device=dev)
for G in (1, 2, 4, 8):
    W = Wl * G
    sw = th.softmax(th.randn(W, R, generator=g, device=dev), -1).contiguous()
    gathered = th.randn(G, 2, B, Wl, A, R, generator=g, device=dev)
    def step():
        ops.envelope_main_forward(ctx, po, obs, sw[:Wl].contiguous())
        ops.envelope_update_shard(ctx, po, grads, obs, act, rew, done, sw, 0, Wl, gathered[0, 0], gathered[0, 1], gamma=0.99,
                                  main_forward_done=True, slab_parts=G)
    for _ in range(5):
        step()
    th.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        step()
    th.cuda.synchronize()
    print(f"G={G} (W_total={W}): main forward + stage B = {(time.perf_counter() - t) / 50 * 1e3:.3f} ms")
