"""Build libmorl_hip.so (hipcc, gfx950 only) in-tree under ``morl-baselines_amd/lib/``.

    python morl-baselines_amd/build.py            # or: __graft_entry__.build()

hipcc cross-compiles without a GPU.  The library is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmorl_hip.so")
SOURCES = ["morl_hip.hip", "morl_ac.hip", "morl_comm.hip"]
HEADERS = ["morl_device.h", "morl_host.h", "gemm_f32.h", "envelope_kernels.h", "mlp_chain.h", "mlp_chain2.h", "mlp_chain16.h", "mlp_chain4.h", "mlp_chain_bf.h", "mlp_chain_bfn.h", "mlp_chain_bf2.h", "mlp_chain_bf_roll.h", "dw_tiles.h", "dw_bf.h", "optim_kernels.h",
           "replay_kernels.h", "pareto_kernels.h", "metrics_kernels.h", "ac_kernels.h", "gemm_wave.h", "gpi_kernels.h", "ens_kernels.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=/path/to/hipcc)")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "morl_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library for gfx950; returns its path."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            # rounding contract: a*b+c is never fused behind our back (the envelope scalarisation must round every
            # product and sum separately, as torch's einsum does); FMAs are written explicitly as fmaf()
            "-ffp-contract=off",
            # A kernel reads its by-value argument block in place (scalar loads from the kernarg segment) only if instcombine can
            # prove the block's local copy is never written -- a walk over the copy's users that gives up after 300 of them.  The
            # split-bf16 chain kernels (six instantiations of the chain body over one argument block) are past that: without this
            # the block is copied to SCRATCH memory at kernel entry and read from there (1.1 KB per work-item, 1.8 x the time).
            # tests/test_build_flags.py holds the kernels to zero scratch.
            "-mllvm", "-instcombine-max-copied-from-constant-users=4000",
            "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    if os.environ.get("MORL_BF_PROF"):     # development build: phase stamps in the split-bf16 chain kernels (mlp_chain_bf.h)
        base.append("-DBF_PROF")
    if os.environ.get("MORL_C16_PROF"):    # development build: phase stamps in the 16-row chain kernel (mlp_chain16.h)
        base.append("-DC16_PROF")
    if os.environ.get("MORL_C4_PROF"):     # development build: wall-clock stamps in the 8-row chain kernel (mlp_chain4.h), printed by the library
        base.append("-DC4_PROF")
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "morl_hip.h")]
    jobs = []
    for src in SOURCES:                     # translation units compile concurrently, unchanged ones are kept
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps + [path])):
            jobs.append((obj, None))
            continue
        cmd = base + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for obj, proc in jobs:
        if proc is not None:
            out, _ = proc.communicate()
            if proc.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + out)
    cmd = base + ["-shared"] + [obj for obj, _ in jobs] + ["-ldl", "-o", LIB_PATH + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
