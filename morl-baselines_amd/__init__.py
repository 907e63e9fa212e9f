"""morl-baselines_amd: MI355X-native (gfx950) hot path of the morl-baselines TD update.

Host code is Python on PyTorch-ROCm and mirrors the reference's class API (MOPolicy / MOAgent /
ReplayBuffer / pareto helpers); the arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/morl_hip.h`` (``lib/libmorl_hip.so``).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import native, ops  # noqa: F401
from .native import load_library  # noqa: F401
