import sys, time, torch as th
sys.path.insert(0, "/root/repo")
from morl_baselines_amd import ops
from morl_baselines_amd.native import load_library
lib = load_library(); dev = th.device("cuda:0")
for rows, arch in ((128, (256, 256)), (256, (256, 256)), (128, (256, 256, 256, 256))):
    ctx = ops.QNetContext(23, 2, 1, arch, rows, 1, lib=lib)
    p = th.randn(ctx.n_params, device=dev) * 0.05
    obs, w = th.randn(rows, 23, device=dev), th.rand(rows, 2, device=dev)
    for _ in range(20): ops.qnet_forward_rows(ctx, p, obs, w)
    th.cuda.synchronize(); t = time.perf_counter()
    for _ in range(500): ops.qnet_forward_rows(ctx, p, obs, w)
    th.cuda.synchronize()
    print(f"rows {rows} arch {arch}: transpose + chain forward = {(time.perf_counter() - t) / 500 * 1e6:.1f} us (engine {ctx.engine})")
