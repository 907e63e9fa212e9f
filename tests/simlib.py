"""TEST INFRASTRUCTURE: build / load the host-emulated kernel library (tests/hipsim).

The unmodified kernel sources of morl-baselines_amd/csrc are compiled for x86 with the wave-level
emulator so that kernel logic can be checked against the oracle without a GPU.  Never used by the
product path, bench.py or smoke().
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, "tests", "hipsim")
OUT = os.path.join(SIM_DIR, "_build", "libmorl_hipsim.so")
CSRC = os.path.join(ROOT, "morl-baselines_amd", "csrc")


def _clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if os.path.exists(c):
            return c
    raise RuntimeError("host clang++ (ROCm llvm) not found")


def build_sim(force=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(SIM_DIR, "hipsim.cpp"), os.path.join(SIM_DIR, "hip", "hip_runtime.h"),
        os.path.join(ROOT, "include", "morl_hip.h")]
    fresh = lambda: os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in srcs)
    if not force and fresh():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # one builder at a time (pytest-xdist workers and the spawned ranks of the gloo tests all come through here), and the library
    # appears atomically: a process never loads a half-written file
    import fcntl
    with open(OUT + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return OUT
        tmp = f"{OUT}.{os.getpid()}.tmp"
        cmd = [_clang(), "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
               "-I", SIM_DIR, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
               os.path.join(CSRC, "morl_hip.hip"), os.path.join(CSRC, "morl_ac.hip"), os.path.join(CSRC, "morl_comm.hip"),
               os.path.join(SIM_DIR, "hipsim.cpp"), "-ldl", "-o", tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emulated build failed:\n" + r.stdout + r.stderr)
        os.replace(tmp, OUT)
    return OUT


def load_sim():
    import morl_baselines_amd.native as native
    return native.NativeLib(build_sim())
