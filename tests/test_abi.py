"""The C-ABI shared library loads on a GPU-less host and exports every symbol include/morl_hip.h declares
(no compute calls here); the ctypes binding covers exactly that set; the product path refuses CPU tensors."""
import ctypes
import os
import re

import pytest
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "morl_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(morl_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported_by_the_gfx950_library():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "morl-baselines_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    path = b.build_library()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/morl_hip.h but not exported"
    lib.morl_abi_version.restype = ctypes.c_int
    from morl_baselines_amd.native import ABI_VERSION
    assert lib.morl_abi_version() == ABI_VERSION == 13
    assert lib.morl_is_device_build() == 1


def test_ctypes_binding_covers_the_header():
    from morl_baselines_amd.native import EXPORTED_SYMBOLS
    assert sorted(EXPORTED_SYMBOLS) == declared_symbols()


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (status codes + morl_last_error)."""
    from morl_baselines_amd.native import load_library, make_net_desc
    lib = load_library()
    d = make_net_desc(32, 3, 6, [256, 256, 256, 256])
    assert lib.param_count(d) == 211218                      # P = 211 218
    d.dims[0] = 99
    h = ctypes.c_void_p()
    rc = lib.lib.morl_ctx_create(ctypes.byref(h), ctypes.byref(d), 256, 64)
    assert rc == -1 and b"dims[0]" in lib.lib.morl_last_error()
    assert lib.lib.morl_polyak(None, None, 1.0, 10, None) == -1
    assert lib.lib.morl_pareto_mask(None, 5, 99, 1, None, None) == -1


@pytest.mark.skipif(th.cuda.is_available(), reason="CPU-only check")
def test_product_path_fails_loudly_on_cpu_tensors():
    import morl_baselines_amd.ops as ops
    from morl_baselines_amd.native import load_library
    lib = load_library()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pareto_mask(lib, th.zeros(4, 2, dtype=th.float64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.polyak(lib, th.zeros(4), th.zeros(4), 0.5)
