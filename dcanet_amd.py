"""Import shim: `import dcanet_amd` loads the package directory
`cost-volume-aggregation-in-stereo-matching-revisited_amd/` (whose name is not a Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "cost-volume-aggregation-in-stereo-matching-revisited_amd")
_spec = importlib.util.spec_from_file_location("dcanet_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dcanet_amd"] = _mod
_spec.loader.exec_module(_mod)
