"""Makes `dcanet_amd` importable when this directory is used as the top-level `models` package
(drop-in for the reference's `from models.gwcnet_dca_g import *`)."""
import importlib.util
import os
import sys


def ensure():
    if "dcanet_amd" in sys.modules:
        return sys.modules["dcanet_amd"]
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dcanet_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dcanet_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
