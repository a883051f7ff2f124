"""Import shim: the package directory is `meshgraphnets.jl_amd/` (a dot in the name cannot be written in
an `import` statement), so it is loaded by path and registered as `mgn_amd`."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_here, "meshgraphnets.jl_amd")
_spec = importlib.util.spec_from_file_location("mgn_amd", os.path.join(_pkg, "__init__.py"),
                                               submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mgn_amd"] = _mod
_spec.loader.exec_module(_mod)
