"""Import shim: the package directory is named `signaloperators.jl_amd` (the repo's
required layout), which is not a valid Python identifier, so it is loaded by path and
registered as `sigops_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "signaloperators.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "sigops_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sigops_amd"] = _mod
_spec.loader.exec_module(_mod)
