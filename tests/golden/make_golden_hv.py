"""Pins the hypervolume to the reference's own implementation -- ``pymoo.indicators.hv.HV`` through
``morl_baselines.common.performance_indicators.hypervolume`` (``performance_indicators.py:15-25``) -- WHEREVER pymoo is
importable, and says so and exits 0 where it is not (this image: pymoo >= 0.6.0 is a dependency of the reference,
``pyproject.toml:31``, that is absent and cannot be installed without a network).

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_hv.py      # writes tests/golden/hv.npz when it can

``tests/test_metrics.py::test_hypervolume_pinned_to_pymoo`` holds the oracle (oracle/metrics_oracle.py) and the device kernel
(``morl_hypervolume``) to the committed file when it exists, and computes the pymoo values live when pymoo is importable in
the test process even if the file was never written -- so the pin lands the first time any environment has pymoo.
Until then the hypervolume is "parity unpinned" against pymoo (oracle pinned to closed forms / the 2-D sweep / Monte-Carlo and
to the literature's 1155 for the Deep-Sea-Treasure front).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "hv.npz")

# (objectives, points, seed): fronts a MORL run produces (a few to a few hundred value vectors, 2-4 objectives) + edge cases
CASES = [(2, 7, 0), (2, 100, 1), (3, 40, 2), (3, 200, 3), (4, 60, 4), (2, 1, 5), (3, 3, 6)]


def case_inputs(R, N, seed):
    """Seeded value vectors with duplicates, dominated points and points that do not dominate the reference point."""
    rng = np.random.default_rng(4000 + seed)
    pts = rng.uniform(-1.0, 3.0, (N, R))
    if N >= 7:
        pts[3] = pts[1]                         # an exact duplicate
        pts[5] = pts[2] - 0.25                  # a dominated point
        pts[6, 0] = -2.0                        # below the reference point in one objective: contributes nothing
    ref = np.full(R, -1.25)
    return ref, pts


def dst_front():
    """The Deep-Sea-Treasure Pareto front (Vamplew et al. 2011) and the reference point the MORL literature uses (HV = 1155)."""
    front = [(1, -1), (2, -3), (3, -5), (5, -7), (8, -8), (16, -9), (24, -13), (50, -14), (74, -17), (124, -19)]
    return np.array([0.0, -25.0]), np.array(front, dtype=np.float64)


def pymoo_available() -> bool:
    try:
        import pymoo.indicators.hv  # noqa: F401
        return getattr(pymoo.indicators.hv, "__file__", None) is not None      # (oracle/ref_harness.py's stub has no file)
    except Exception:
        return False


def reference_hypervolume():
    """``hypervolume`` of the unmodified reference when its tree is present, else the same one-liner on pymoo itself."""
    ref_root = "/root/reference"
    if os.path.isdir(os.path.join(ref_root, "morl_baselines")):
        src = open(os.path.join(ref_root, "morl_baselines", "common", "performance_indicators.py")).read()
        ns = {}
        # the module's top imports are numpy / pymoo only; exec it as it is (unmodified), take its function
        exec(compile(src, "performance_indicators.py", "exec"), ns)
        return ns["hypervolume"]
    from pymoo.indicators.hv import HV
    return lambda ref_point, points: HV(ref_point=ref_point * -1)(np.array(points) * -1)     # performance_indicators.py:25


def compute():
    hv = reference_hypervolume()
    out = {}
    for k, (R, N, seed) in enumerate(CASES):
        ref, pts = case_inputs(R, N, seed)
        out[f"hv_{k}"] = np.float64(hv(ref, list(pts)))
    ref, pts = dst_front()
    out["hv_dst"] = np.float64(hv(ref, list(pts)))
    return out


def main():
    if not pymoo_available():
        print("pymoo is not importable here: the hypervolume stays UNPINNED against pymoo (nothing written)")
        return 0
    out = compute()
    np.savez(OUT, **out)
    print({k: float(v) for k, v in out.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
