"""TEST INFRASTRUCTURE: known-answer fronts of the reference's own pruning tests, as DATA.

Imports the reference's ``tests/test_pruning.py`` from /root/reference (build container only) and calls ITS generators with ITS
seed and sizes -- ``test_small_pf`` (100 + 500 points x 2 objectives, ``tests/test_pruning.py:71-84``) and ``test_large_pf``
(1 000 + 5 000 x 4, ``:98-110``) -- then runs the unmodified ``filter_pareto_dominated`` / ``get_non_pareto_dominated_inds``
(``common/pareto.py:34-73``) on the stacked set.  Committed: the stacked points (float64), the number of non-dominated points the
generator planted, the reference's mask for remove_duplicates True / False.  No source of the reference is stored.

    python tests/golden/make_golden_pruning.py
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ref_harness as rh  # noqa: E402

CASES = {"small_pf": (100, 500, 2), "large_pf": (1000, 5000, 4)}     # the reference tests' own sizes, seed 0 (TestPruning.test_seed)


def main():
    ref = rh.import_reference()
    spec = importlib.util.spec_from_file_location("_ref_test_pruning", os.path.join(rh.REFERENCE_ROOT, "tests", "test_pruning.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, (n_nd, n_d, dims) in CASES.items():
        rng = np.random.default_rng(mod.TestPruning.test_seed)
        nd, d = mod.generate_known_front(n_nd, n_d, dims=dims, decimals=None, min_val=0, max_val=10, rng=rng)
        pts = np.vstack((nd, d))
        kept = ref.pareto.filter_pareto_dominated(pts)
        assert {tuple(v) for v in kept} == {tuple(v) for v in nd}, name           # the reference's own assertion holds here
        out[f"{name}__points"] = pts
        out[f"{name}__n_nd"] = np.int64(n_nd)
        for rd in (True, False):
            out[f"{name}__mask_rd{int(rd)}"] = np.atleast_1d(ref.pareto.get_non_pareto_dominated_inds(pts, remove_duplicates=rd)).astype(np.uint8)
        print(name, pts.shape, int(out[f"{name}__mask_rd1"].sum()), "kept")
    np.savez_compressed(os.path.join(HERE, "pruning_known_fronts.npz"), **out)


if __name__ == "__main__":
    main()
