"""Importable alias of the package directory `dot-socp_amd/` (a hyphen is not allowed in a
Python identifier): `import dotsocp_amd` loads that package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dot-socp_amd")
_spec = importlib.util.spec_from_file_location(
    "dotsocp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dotsocp_amd"] = _mod
_spec.loader.exec_module(_mod)
