"""Import alias for the package directory ``morl-baselines_amd/`` (a hyphen is not importable by name).

``import morl_baselines_amd`` (and ``morl_baselines_amd.envelope`` etc.) resolves to that directory.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "morl-baselines_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
