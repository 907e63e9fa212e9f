"""pytest configuration: registers the ``gpu`` marker and puts the repo root / oracle on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The library evaluates the target network lazily only for steps of >= 8 192 TD rows (smaller, latency-bound steps are faster
# eagerly).  The suite's cases are almost all small, so it lowers the threshold to zero: every Envelope step that does NOT ask for
# the whole target slab then runs the lazy pipeline -- the agents, the training traces, and the ``lazy`` legs of the fixture tests
# (tests/test_kernels_parity.py ``debug="lazy"``, tests/test_flagship_golden.py at both full sizes).  A step that asks for
# ``q_target_next`` (``debug=True``: the ``eager`` legs of the same tests) is evaluated eagerly whatever this setting, as are DDQN
# targets, the weight-sharded steps and everything under MORL_LAZY_TARGETS=0.
os.environ.setdefault("MORL_LAZY_MIN_ROWS", "0")
# Same for the split-bf16 chain (csrc/mlp_chain_bf.h: the two online forward passes and the dX backward pass of steps of >= 8 192 TD
# rows on 256-wide networks): with the threshold at zero every fixture / agent whose network qualifies (flagship_b32w8, wide_pick,
# both full-size cases, the shape sweep's all-256 nets) runs it, on the emulator and on the GPU; tests/test_chain_tilings.py re-runs
# the fixtures with MORL_EXACT_F32=1 (every GEMM on the f32-input MFMA).
os.environ.setdefault("MORL_BF_MIN_ROWS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
