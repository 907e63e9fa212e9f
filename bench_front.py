"""Front maintenance on one MI355X: the Pareto prune (``common/pareto.py:34-57`` -> ``morl_pareto_mask``) and the exact
hypervolume / expected utility of the pruned front (``common/performance_indicators.py`` -> ``morl_hypervolume``,
``morl_expected_utility``).  bench.py stays the north-star line; this script measures SURVEY section 8 rows P1 / P2 and
section 8(f) rank 4 with the same conventions (inputs resident in HBM, one JSON line per workload, CPU baseline from the oracle
on a bounded sample).

    python bench_front.py --workload pareto [--n 16384] [--r 3]
    python bench_front.py --workload hv [--n 100] [--r 3]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch as th


# f64 compares a SIMD retires per clock (v_cmp_le_f64 at 4 waves per SIMD issuing to rotating scalar destinations: 1.81 cycles per
# wave-instruction = 35.3 lane-compares per clock; tools/probes/cmp64_probe.hip, profiles/r06_cmp64_probe.txt) x 1 024 SIMDs x 2.4 GHz:
# the rate the Pareto kernel's compare instructions are priced against
F64_CMP_LANES_PER_CLK_SIMD = 35.3
PEAK_F64_COMPARES = F64_CMP_LANES_PER_CLK_SIMD * 1024 * 2.4e9


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    th.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pareto", choices=["pareto", "hv"])
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--r", type=int, default=3)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    if not th.cuda.is_available():
        raise SystemExit("bench_front.py needs an MI355X (no CPU fallback exists)")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from morl_baselines_amd import ops
    from morl_baselines_amd import performance_indicators as pi
    from morl_baselines_amd.native import load_library

    lib = load_library()
    dev = th.device("cuda", 0)
    rng = np.random.default_rng(0)
    R = a.r
    if a.workload == "pareto":
        N = a.n or 16384
        # points near a sphere shell: a large non-dominated fraction, the expensive case (no early exit for most rows)
        x = np.abs(rng.standard_normal((N, R)))
        x = x / np.linalg.norm(x, axis=1, keepdims=True) * rng.uniform(0.9, 1.0, (N, 1))
        x[rng.integers(0, N, N // 50)] = x[0]                    # duplicates
        pts = th.from_numpy(x).to(dev)
        sec = timed(lambda: ops.pareto_mask(lib, pts, True), a.steps, a.warmup)
        kept = int(ops.pareto_mask(lib, pts, True).sum().item())
        # the roofline leg: the same launch with its wave-uniform early exits switched off (bit 1 of the flag), so that it executes
        # a KNOWN number of pair tests -- N^2, each R (<=, ==) compare pairs -- and the same mask
        sec_all = timed(lambda: ops.pareto_mask(lib, pts, 3), a.steps, a.warmup)
        assert bool((ops.pareto_mask(lib, pts, 3) == ops.pareto_mask(lib, pts, True)).all())
        pair_tests = float(N) * N
        cmp_rate = 2.0 * R * pair_tests / sec_all
        out = {"metric": "Pareto prune candidates/sec", "value": N / sec, "unit": "candidates/s", "n_gpus": 1,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"get_non_pareto_dominated_inds on {N} candidates x {R} objectives (float64), "
                                      f"{kept} non-dominated, 2% duplicates"},
               "roofline": {"bound": "compare (f64 VALU)", "kernel": "pareto_mask_kernel",
                            "achieved": cmp_rate / 1e12, "peak": PEAK_F64_COMPARES / 1e12, "unit": "T f64 compares/s",
                            "frac": cmp_rate / PEAK_F64_COMPARES, "traffic": None,
                            "executed": "2 R N^2 v_cmp_*_f64 lane-operations per launch with the early exits switched off "
                                        f"({sec_all * 1e3:.4f} ms; {sec * 1e3:.4f} ms with them: ms_per_step / value)",
                            "peak_source": f"{F64_CMP_LANES_PER_CLK_SIMD:g} f64 compares per clock and SIMD (tools/probes/cmp64_probe.hip, "
                                           "profiles/r06_cmp64_probe.txt) x 1 024 SIMDs x 2.4 GHz",
                            "early_exit_speedup": sec_all / sec,
                            "algorithmic_bytes_per_launch": N * R * 8 + N,
                            "note": "N*R*8 B in, N B out: every workgroup re-reads the candidate set from L2 (N*R*8 B per "
                                    "256 candidates); compare-bound, not HBM-bound"}}
        if not a.no_cpu_baseline:
            import envelope_oracle as orc
            n_cpu = min(N, 4096)                                 # the oracle is O(N^2) python/numpy: bounded sample
            t1 = time.perf_counter()
            orc.pareto_mask(x[:n_cpu], True)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": n_cpu / dt, "unit": "candidates/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle/envelope_oracle.py::pareto_mask on the first {n_cpu} candidates "
                                             f"({dt:.2f} s; cost grows ~N^2)"}
    else:
        N = a.n or 100
        front = rng.uniform(0.1, 1.0, (N, R))
        pts = th.from_numpy(front).to(dev)
        ref = th.zeros(R, dtype=th.float64, device=dev)
        sec = timed(lambda: pi.hypervolume_device(ref, pts, lib), a.steps, a.warmup)
        tests = float(N) ** (R - 1) * N
        out = {"metric": "exact hypervolume evaluations/sec", "value": 1.0 / sec, "unit": "fronts/s", "n_gpus": 1,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"hypervolume of {N} points x {R} objectives (float64), slab decomposition: "
                                      f"{int(float(N) ** (R - 1))} boxes x {N} point tests"},
               "roofline": {"bound": "compare (f64 VALU)", "kernel": "hv_boxes_kernel", "achieved": tests * (R - 1) / sec / 1e12,
                            "peak": None, "unit": "T coordinate-compares/s", "frac": None, "traffic": None,
                            "note": "3 launches (sort, boxes, finish); below ~1e6 point tests the call is dispatch-latency-bound"}}
        if not a.no_cpu_baseline:
            import metrics_oracle as mo
            t1 = time.perf_counter()
            want = mo.hypervolume(np.zeros(R), list(front))
            dt = time.perf_counter() - t1
            got = float(pi.hypervolume_device(ref, pts, lib).item())
            out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "fronts/s", "cores": 1, "kind": "port",
                                   "sample": f"one oracle/metrics_oracle.py::hypervolume call (HSO recursion, {dt:.3f} s); "
                                             f"relative difference to the device value {abs(got - want) / want:.1e}"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
