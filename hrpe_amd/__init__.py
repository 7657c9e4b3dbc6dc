"""Import shim: the package directory is ``holistic-robot-pose-estimation_amd`` (not a valid Python
identifier); this module loads it under the importable name ``hrpe_amd``."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "holistic-robot-pose-estimation_amd")
_spec = importlib.util.spec_from_file_location("hrpe_amd", os.path.join(_root, "__init__.py"),
                                               submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hrpe_amd"] = _mod
_spec.loader.exec_module(_mod)
