"""Golden vectors for the front metrics the reference computes itself (sparsity, expected_utility, maximum_utility_loss,
cardinality -- ``common/performance_indicators.py:41-130``), produced by the UNMODIFIED reference functions with pymoo
stubbed out for the import (oracle/ref_harness.py).  hypervolume / igd call pymoo, which this image does not have: no golden
value exists for them (see oracle/metrics_oracle.py, "parity unpinned").

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_metrics.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_harness as rh  # noqa: E402

CASES = [(2, 7, 11, 0), (3, 40, 50, 1), (4, 100, 100, 2), (2, 1, 5, 3), (6, 33, 20, 4)]      # R, N points, M weights, seed


def case_inputs(R, N, M, seed):
    rng = np.random.default_rng(seed)
    front = rng.uniform(-1.0, 3.0, (N, R))
    weights = rng.dirichlet(np.ones(R), M)
    ref_set = rng.uniform(-1.0, 3.5, (N + 3, R))
    return front, weights, ref_set


def main():
    rh.install_stubs()
    rh.import_reference()
    from morl_baselines.common import performance_indicators as pi

    out = {}
    for k, (R, N, M, seed) in enumerate(CASES):
        front, weights, ref_set = case_inputs(R, N, M, seed)
        fl, wl, rl = list(front), list(weights), list(ref_set)
        out[f"sparsity_{k}"] = np.float64(pi.sparsity(fl))
        out[f"eum_{k}"] = np.float64(pi.expected_utility(fl, wl))
        out[f"mul_{k}"] = np.float64(pi.maximum_utility_loss(fl, rl, weights))
        out[f"card_{k}"] = np.float64(pi.cardinality(fl))
    np.savez(os.path.join(HERE, "metrics.npz"), **out)
    print({k: float(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
