"""Import shim: the package directory is named `ch-bin_amd/` (not a valid Python identifier),
so `import chbin_amd` loads that directory as the package `chbin_amd`."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ch-bin_amd")
_spec = importlib.util.spec_from_file_location(
    "chbin_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["chbin_amd"] = _mod
_spec.loader.exec_module(_mod)
