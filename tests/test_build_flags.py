"""The hot kernels must not touch scratch memory.  hipcc copies a kernel's by-value argument block to scratch when instcombine
cannot prove the block's local copy read-only (its walk over the users stops at 300: ``build.py`` raises the limit); a dynamically
indexed local array lands there too.  Either costs the chain kernels most of their speed without failing a single parity test --
so the build is checked: compile the main translation unit to assembly with the library's flags and read the kernels' metadata."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HOT = ("mlp_chain_bf_kernel", "mlp_chain_bf32_kernel", "mlp_chain_bf2_kernel", "mlp_chain_bf_roll_kernel", "mlp_chain_bf_pw_kernel", "mlp_chain_bf32_pw_kernel", "mlp_chain_bf_fwd_pw_kernel", "mlp_chain_bf32_fwd_pw_kernel", "mlp_chain_bfn_kernel", "mlp_chain_bfn16_kernel", "dw_bf_kernel", "mlp_chain4_kernel", "mlp_chain16_kernel", "dw_tiles_kernel",
       "step_prologue_kernel", "clip_adam_kernel", "grad_reduce_ranges_kernel")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_hot_kernels_use_no_scratch_memory(tmp_path):
    import importlib
    build = importlib.import_module("morl-baselines_amd.build")
    src = open(os.path.join(build.ROOT, "morl-baselines_amd", "build.py")).read()
    flags = re.search(r'base = \[_hipcc\(\), (.*?)\]\n', src, flags=re.S)
    assert flags is not None and "-instcombine-max-copied-from-constant-users" in flags.group(1)
    out = tmp_path / "morl_hip.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
           "-instcombine-max-copied-from-constant-users=4000", "-I", os.path.join(ROOT, "include"), "-I", build.CSRC, "-S",
           "--cuda-device-only", os.path.join(build.CSRC, "morl_hip.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    text = out.read_text()
    seen = {}
    for m in re.finditer(r"^(_Z\w+):.*?; ScratchSize: (\d+)", text, flags=re.S | re.M):
        for k in HOT:
            if k in m.group(1) and m.group(1) not in seen:
                seen[m.group(1)] = (k, int(m.group(2)))
    assert {k for k, _ in seen.values()} == set(HOT), sorted(seen)
    bad = {n: sz for n, (k, sz) in seen.items() if sz > 0}
    assert not bad, bad
