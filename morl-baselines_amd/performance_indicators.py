"""Front metrics (mirror of ``common/performance_indicators.py``): the two that scale with the archive -- hypervolume and
expected utility -- run on the device (``morl_hypervolume``: exact slab decomposition, ``morl_expected_utility``;
``csrc/metrics_kernels.h``); sparsity / cardinality / maximum utility loss / IGD are a few numpy lines as in the reference.

The reference's ``hypervolume`` and ``igd`` call pymoo; these functions need nothing beyond the HIP library, so fronts can
be scored on machines without pymoo (the GPU box).  Same signatures, same conventions (maximisation, ``ref_point`` the
worst corner)."""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch as th

from .native import NativeLib, load_library


PRUNE_ABOVE = 256     # fronts with more points are Pareto-pruned on the device before the hypervolume kernel


def _dev(lib: NativeLib, device=None) -> th.device:
    if device is not None:
        return th.device(device)
    return th.device("cuda:0") if lib.is_device_build else th.device("cpu")


def _f64(x, dev) -> th.Tensor:
    return th.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64))).to(dev).contiguous()


def _workspace(lib: NativeLib, N: int, R: int, M: int, dev) -> th.Tensor:
    n = int(lib.lib.morl_metrics_workspace_doubles(N, R))
    if n < 0:
        raise ValueError(f"bad front shape N={N} R={R}")
    return th.empty(max(n, (M + 3) // 4, 1), dtype=th.float64, device=dev)


def hypervolume_device(ref_point: th.Tensor, points: th.Tensor, lib: Optional[NativeLib] = None) -> th.Tensor:
    """Hypervolume of device float64 ``points`` (N, R) w.r.t. ``ref_point`` (R,) as a device float64 scalar (no sync)."""
    lib = lib or load_library()
    N, R = points.shape
    out = th.empty(1, dtype=th.float64, device=points.device)
    ws = _workspace(lib, N, R, 0, points.device)
    lib.check_device(points, ref_point)
    lib.check(lib.lib.morl_hypervolume(points.data_ptr() if N else None, N, R, ref_point.data_ptr(), ws.data_ptr(),
                                       out.data_ptr(), lib.stream_of(out)))
    return out[0]


def hypervolume(ref_point: np.ndarray, points: List[np.ndarray], lib: Optional[NativeLib] = None, device=None) -> float:
    """``performance_indicators.py:15-25``."""
    lib = lib or load_library()
    dev = _dev(lib, device)
    ref = _f64(ref_point, dev).reshape(-1)
    pts = _f64(np.array(points), dev).reshape(-1, ref.numel())
    if pts.shape[0] > PRUNE_ABOVE:
        # a large archive (the reference's pymoo HV takes any N): dominated and duplicate points add nothing to the dominated
        # volume, so the device Pareto prune shrinks the front first; what is left may still exceed the LDS-staged size
        # (it is then read in place) but not the kernel's test budget, which is refused loudly
        from . import ops
        keep = ops.pareto_mask(lib, pts, remove_duplicates=True).bool()
        pts = pts[keep].contiguous()
    return float(hypervolume_device(ref, pts, lib).item())


def expected_utility(front: List[np.ndarray], weights_set: List[np.ndarray], utility: Callable = np.dot,
                     lib: Optional[NativeLib] = None, device=None) -> float:
    """``performance_indicators.py:71-91``.  The dot-product utility runs on the device; any other utility is evaluated
    with the reference's host loop."""
    if utility is not np.dot:
        maxs = [np.max(np.array([utility(w, p) for p in front])) for w in weights_set]
        return np.mean(np.array(maxs), axis=0)
    lib = lib or load_library()
    dev = _dev(lib, device)
    f = _f64(np.array(front), dev)
    f = f.reshape(len(front), -1)
    w = _f64(np.array(weights_set), dev).reshape(-1, f.shape[1])
    out = th.empty(1, dtype=th.float64, device=dev)
    ws = _workspace(lib, f.shape[0], f.shape[1], w.shape[0], dev)
    lib.check_device(f, w)
    lib.check(lib.lib.morl_expected_utility(f.data_ptr(), f.shape[0], f.shape[1], w.data_ptr(), w.shape[0], ws.data_ptr(),
                                            out.data_ptr(), lib.stream_of(out)))
    return float(out.item())


def sparsity(front: List[np.ndarray]) -> float:
    """``performance_indicators.py:41-68``: mean squared gap between neighbours, per objective."""
    if len(front) < 2:
        return 0.0
    f = np.array(front)
    total = 0.0
    for dim in range(f.shape[1]):
        o = np.sort(f.T[dim].copy())
        for i in range(1, len(o)):
            total += np.square(o[i] - o[i - 1])
    return total / (len(f) - 1)


def cardinality(front: List[np.ndarray]) -> float:
    """``performance_indicators.py:94-105``."""
    return len(front)


def maximum_utility_loss(front: List[np.ndarray], reference_set: List[np.ndarray], weights_set: np.ndarray,
                         utility: Callable = np.dot) -> float:
    """``performance_indicators.py:108-130``."""
    best_ref = [np.max([utility(w, p) for p in reference_set]) for w in weights_set]
    best = [np.max([utility(w, p) for p in front]) for w in weights_set]
    return np.max([best_ref[i] - best[i] for i in range(len(best))])


def igd(known_front: List[np.ndarray], current_estimate: List[np.ndarray]) -> float:
    """``performance_indicators.py:28-38`` (pymoo ``IGD``): mean distance from each known-front point to its nearest
    estimate."""
    z = np.asarray(known_front, dtype=np.float64)
    a = np.asarray(current_estimate, dtype=np.float64)
    d = np.sqrt(((z[:, None, :] - a[None, :, :]) ** 2).sum(-1))
    return float(d.min(axis=1).mean())
