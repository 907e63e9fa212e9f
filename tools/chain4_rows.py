"""Development probe: the 8-row-tile forward chain (mlp_chain4_kernel) at the flagship network, a few row counts, HIP-event timed."""
import sys, torch as th
sys.path.insert(0, "/root/repo")
from morl_baselines_amd import ops
from morl_baselines_amd.native import load_library
lib = load_library(); dev = th.device("cuda:0")
for rows in (8, 512, 1400, 2048):
    ctx = ops.QNetContext(32, 3, 6, (256, 256, 256, 256), rows, 1, lib=lib)
    p = th.randn(ctx.n_params, device=dev) * 0.05
    obs, w = th.randn(rows, 32, device=dev), th.rand(rows, 3, device=dev)
    for _ in range(20): ops.qnet_forward_rows(ctx, p, obs, w)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): ops.qnet_forward_rows(ctx, p, obs, w)
    e1.record(); th.cuda.synchronize()
    print(f"rows {rows}: {e0.elapsed_time(e1) / 300 * 1e3:.1f} us per call (back to back)")
