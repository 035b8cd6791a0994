"""Import shim: the package directory is `clearsky.jl_amd/` (not a valid Python identifier), so this module loads it
under the importable name `clearsky_jl_amd`.  `import clearsky_jl_amd as cs` works from the repo root."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clearsky.jl_amd")
_spec = importlib.util.spec_from_file_location("clearsky_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["clearsky_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
