"""Where the large tiles take over from the 8-row tiles for the lazily evaluated target rows: the flagship-shape step with K
distinct selected pairs per transition (K = 2 .. 64 -> 512 .. 16 384 compact rows), timed with the target launch forced onto the
small tiles (MORL_LAZY_BIG_ROWS above every count) and onto the large ones (MORL_LAZY_BIG_ROWS=1), and evaluated eagerly.  Run on the
GPU box: python tools/lazy_sweep.py > gpurun_out/.../lazy_sweep.json"""
import json
import os
import sys

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]
import morl_baselines_amd.ops as ops                      # noqa: E402
from morl_baselines_amd.native import load_library        # noqa: E402
import test_lazy_adaptive as T                            # noqa: E402


def main():
    lib, dev = load_library(), th.device("cuda:0")
    B, W, D, A, arch = 256, 64, 32, 6, (256, 256, 256, 256)
    out = []
    for K in (2, 4, 8, 12, 16, 24, 32, 48, 64):
        inp = T.crafted_inputs(B, W, D, A, arch)
        dirs = T.octant_directions(K)
        inp["sampled_w"] = dirs[np.arange(W) % K].copy()          # row i selects the FIRST weight with its direction: j* = i % K
        rec = {"pairs_per_transition": K}
        for name, lazy, big in (("small_tiles", 1, str(1 << 30)), ("large_tiles", 1, "1"), ("eager", 0, "1")):
            os.environ["MORL_LAZY_BIG_ROWS"] = big
            ctx = ops.QNetContext(D, 3, A, arch, B, W, lib=lib)
            ctx.set_lazy_targets(lazy)
            po, pt = T.flat(inp["online"]).to(dev), T.flat(inp["target"]).to(dev)
            g, m, v = th.zeros_like(po), th.zeros_like(po), th.zeros_like(po)
            args = T.step_args(inp, dev)
            best = 1e9
            for rep in range(3):
                for k in range(10):
                    ops.envelope_update(ctx, po, pt, g, m, v, *args, gamma=0.99, lr=1e-9, adam_step=k + 1, max_grad_norm=1.0)
                e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
                th.cuda.synchronize()
                e0.record()
                for k in range(40):
                    ops.envelope_update(ctx, po, pt, g, m, v, *args, gamma=0.99, lr=1e-9, adam_step=11 + k, max_grad_norm=1.0)
                e1.record()
                th.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 40)
            rec[name + "_ms"] = round(best, 5)
            rec["rows"] = max(rec.get("rows", 0), ctx.lazy_target_rows(po))
            rec[name + "_bits"] = ctx.last_step_bf16()
            ctx.close()
        out.append(rec)
        print(rec, file=sys.stderr, flush=True)
    print(json.dumps({"what": "flagship-shape step (256 x 64 x 3), K distinct selected pairs per transition; ms per step (best of 3 x 40)",
                      "sweep": out}))


if __name__ == "__main__":
    main()
