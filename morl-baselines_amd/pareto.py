"""Pareto utilities with the dominance tests on the device (mirror of ``common/pareto.py``).

``get_non_pareto_dominated_inds`` / ``filter_pareto_dominated`` / ``ParetoArchive`` keep the reference's signatures
and return types (numpy in, numpy out); the O(N^2 R) float64 comparisons run in ``morl_pareto_mask``
(``csrc/pareto_kernels.h``).  The convex-hull variant stays on SciPy (``pareto.py:76-93``, out of the hot path).
"""
from __future__ import annotations

from copy import deepcopy
from typing import List, Optional, Union

import numpy as np
import torch as th

from . import ops
from .native import NativeLib, load_library

_device: Optional[th.device] = None


def _dev() -> th.device:
    global _device
    if _device is None:
        _device = th.device("cuda:0")
    return _device


def get_non_pareto_dominated_inds(candidates: Union[np.ndarray, List], remove_duplicates: bool = True,
                                  lib: Optional[NativeLib] = None, device=None) -> np.ndarray:
    """Boolean keep-mask, bit-exact with ``pareto.py:34-57`` (float64 comparisons, first duplicate kept)."""
    lib = lib or load_library()
    c = np.ascontiguousarray(np.array(candidates), dtype=np.float64)
    if c.ndim != 2:
        raise ValueError("candidates must be a 2-D array of objective vectors")
    if c.shape[0] == 0:
        return np.zeros(0, dtype=bool)
    dev = th.device(device) if device is not None else (_dev() if lib.is_device_build else th.device("cpu"))
    mask = ops.pareto_mask(lib, th.from_numpy(c).to(dev), remove_duplicates)
    return mask.cpu().numpy().astype(bool)


def filter_pareto_dominated(candidates: Union[np.ndarray, List], remove_duplicates: bool = True,
                            lib: Optional[NativeLib] = None, device=None) -> np.ndarray:
    """``pareto.py:60-73``."""
    candidates = np.array(candidates)
    if len(candidates) < 2:
        return candidates
    return candidates[get_non_pareto_dominated_inds(candidates, remove_duplicates=remove_duplicates, lib=lib,
                                                    device=device)]


def filter_convex_dominated(candidates: Union[np.ndarray, List], lib: Optional[NativeLib] = None, device=None):
    """``pareto.py:76-93``: QuickHull (SciPy, host) then the device Pareto filter."""
    from scipy.spatial import ConvexHull

    candidates = np.array(candidates)
    ccs = candidates[ConvexHull(candidates).vertices] if len(candidates) > 2 else candidates
    return filter_pareto_dominated(ccs, lib=lib, device=device)


class ParetoArchive:
    """``pareto.py:140-175``: keeps the non-dominated evaluations and (deep copies of) their individuals."""

    def __init__(self, convex_hull: bool = False, lib: Optional[NativeLib] = None, device=None):
        self.convex_hull = convex_hull
        self.individuals: list = []
        self.evaluations: List[np.ndarray] = []
        self._lib, self._device = lib, device

    def add(self, candidate, evaluation: np.ndarray):
        self.evaluations.append(evaluation)
        self.individuals.append(deepcopy(candidate))
        if self.convex_hull:
            nd = {tuple(x) for x in filter_convex_dominated(self.evaluations, lib=self._lib, device=self._device)}
        else:
            nd = {tuple(x) for x in filter_pareto_dominated(self.evaluations, lib=self._lib, device=self._device)}
        keep_e, keep_t, keep_i = [], [], []
        for e, ind in zip(self.evaluations, self.individuals):
            if tuple(e) in nd and tuple(e) not in keep_t:
                keep_e.append(e)
                keep_t.append(tuple(e))
                keep_i.append(ind)
        self.evaluations, self.individuals = keep_e, keep_i
