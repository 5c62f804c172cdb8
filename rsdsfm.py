"""Import shim: exposes the package directory `rs-aware-differential-sfm_amd/` (not a valid Python
identifier) as the importable module `rsdsfm`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rs-aware-differential-sfm_amd")
_spec = importlib.util.spec_from_file_location("rsdsfm", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rsdsfm"] = _mod
_spec.loader.exec_module(_mod)
