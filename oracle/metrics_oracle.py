"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement of the front metrics of ``common/performance_indicators.py``.

PARITY UNPINNED for ``hypervolume``: the reference computes it with pymoo (``pymoo >= 0.6.0``, pyproject.toml:31:
``HV(ref_point=ref_point * -1)(np.array(points) * -1)``, performance_indicators.py:15-25), which is absent from this image
and from /root/reference, and the reference's tests hold no hypervolume golden value.  The quantity itself is unambiguous
-- the Lebesgue measure of the region dominated by the points and dominating the reference point; pymoo's implementation is
the exact recursive dimension-sweep of Fonseca et al. -- so the restatement below (hypervolume by slicing objectives, HSO)
is pinned to closed-form cases, to the 2-D sweep of tests/momdp.py and to a Monte-Carlo estimate in tests/test_metrics.py.
``sparsity`` / ``expected_utility`` / ``cardinality`` / ``maximum_utility_loss`` / ``igd`` follow the reference line by
line and are checked against the imported reference functions where those import without pymoo (the generator script
tests/golden/make_golden_metrics.py stubs pymoo to import the module; HV / IGD themselves are not callable that way).
"""
from __future__ import annotations

from typing import Callable, List

import numpy as np


def hypervolume(ref_point: np.ndarray, points: List[np.ndarray]) -> float:
    """``performance_indicators.py:15-25``: volume dominated by ``points`` and dominating ``ref_point`` (maximisation).
    Hypervolume by slicing objectives: sweep the last objective from the best point down, the slab between two
    consecutive values has the (R-1)-dimensional hypervolume of the points at or above it as cross-section."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, len(ref_point))
    ref = np.asarray(ref_point, dtype=np.float64)
    pts = pts[(pts > ref).all(axis=1)]           # a point that is not strictly better than ref dominates nothing of the box
    return _hso(pts, ref)


def _hso(pts: np.ndarray, ref: np.ndarray) -> float:
    if len(pts) == 0:
        return 0.0
    if pts.shape[1] == 1:
        return float(pts[:, 0].max() - ref[0])
    order = np.argsort(-pts[:, -1], kind="stable")
    pts = pts[order]
    total = 0.0
    for i in range(len(pts)):
        lo = pts[i + 1, -1] if i + 1 < len(pts) else ref[-1]
        depth = pts[i, -1] - lo
        if depth > 0.0:
            total += depth * _hso(pts[:i + 1, :-1], ref[:-1])
    return float(total)


def sparsity(front: List[np.ndarray]) -> float:
    """``performance_indicators.py:41-68``."""
    if len(front) < 2:
        return 0.0
    f = np.array(front)
    s = 0.0
    for dim in range(f.shape[1]):
        o = np.sort(f.T[dim].copy())
        for i in range(1, len(o)):
            s += np.square(o[i] - o[i - 1])
    return s / (len(f) - 1)


def expected_utility(front: List[np.ndarray], weights_set: List[np.ndarray], utility: Callable = np.dot) -> float:
    """``performance_indicators.py:71-91``."""
    maxs = [np.max(np.array([utility(w, p) for p in front])) for w in weights_set]
    return np.mean(np.array(maxs), axis=0)


def cardinality(front: List[np.ndarray]) -> float:
    """``performance_indicators.py:94-105``."""
    return len(front)


def maximum_utility_loss(front, reference_set, weights_set, utility: Callable = np.dot) -> float:
    """``performance_indicators.py:108-130``."""
    max_ref = [np.max(np.array([utility(w, p) for p in reference_set])) for w in weights_set]
    max_front = [np.max(np.array([utility(w, p) for p in front])) for w in weights_set]
    return np.max([a - b for a, b in zip(max_ref, max_front)])


def igd(known_front: List[np.ndarray], current_estimate: List[np.ndarray]) -> float:
    """``performance_indicators.py:28-38`` (pymoo ``IGD``): mean over the known front of the Euclidean distance to the
    nearest estimate point."""
    z = np.asarray(known_front, dtype=np.float64)
    a = np.asarray(current_estimate, dtype=np.float64)
    d = np.sqrt(((z[:, None, :] - a[None, :, :]) ** 2).sum(-1))
    return float(d.min(axis=1).mean())
