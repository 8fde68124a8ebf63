"""Import shim: the package directory is `lt-mapper_amd/` (not a valid Python identifier), so this
module loads it under the importable name `ltmapper_amd`.  `import ltmapper_amd` anywhere with the
repo root on sys.path gives the package."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_here, "lt-mapper_amd")
_spec = importlib.util.spec_from_file_location(
    "ltmapper_amd", os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ltmapper_amd"] = _mod
_spec.loader.exec_module(_mod)
