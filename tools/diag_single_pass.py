"""One forward pass of the flagship network over 16 384 rows, bracketed by HIP events: the persistent chain at 2 workgroups per
CU (32-row halves) and at one (MORL_CHAIN_SLOTS=256, 64-row units) -- what a lazily evaluated step pays for splitting its
two-pass forward launch."""
import os, subprocess, sys
if len(sys.argv) > 1:
    import torch as th
    sys.path.insert(0, "/root/repo")
    from morl_baselines_amd import ops
    from morl_baselines_amd.native import load_library
    lib = load_library(); dev = th.device("cuda:0")
    rows = int(sys.argv[1])
    ctx = ops.QNetContext(7, 3, 6, (256, 256, 256, 256), rows, 1, lib=lib)
    p = th.randn(ctx.n_params, device=dev) * 0.05
    obs, w = th.randn(rows, 7, device=dev), th.rand(rows, 3, device=dev)
    for _ in range(20): ops.qnet_forward_rows(ctx, p, obs, w)
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    th.cuda.synchronize(); e0.record()
    for _ in range(200): ops.qnet_forward_rows(ctx, p, obs, w)
    e1.record(); th.cuda.synchronize()
    print(f"rows {rows} slots {os.environ.get('MORL_CHAIN_SLOTS', 'default')}: shadow copy + one pass = {e0.elapsed_time(e1) / 200 * 1e3:.1f} us")
else:
    for rows in ("16384", "1355", "2048"):
        for slots in (None, "256", "512"):
            env = dict(os.environ)
            if slots: env["MORL_CHAIN_SLOTS"] = slots
            subprocess.run([sys.executable, __file__, rows], env=env)
