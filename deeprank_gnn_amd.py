"""Import alias: the package lives in ``deeprank-gnn_amd/`` (not a valid Python
identifier), this shim registers it under the importable name ``deeprank_gnn_amd``."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deeprank-gnn_amd")
_spec = importlib.util.spec_from_file_location(
    "deeprank_gnn_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules["deeprank_gnn_amd"] = _module
_spec.loader.exec_module(_module)
