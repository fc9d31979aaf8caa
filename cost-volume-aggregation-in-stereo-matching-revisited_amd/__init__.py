"""MI355X-native DCANet cost-volume hot path (gfx950 HIP kernels behind the reference's Python API).

Import as `dcanet_amd` (see the shim at the repository root).  Mirrors the reference module layout:
    dcanet_amd.models.submodule      <- models/submodule.py   (build_gwc_volume, disparity_regression, ...)
    dcanet_amd.models.gwcnet_dca_g   <- models/gwcnet_dca_g.py (GwcNet, GwcNet_G, GwcNet_GC)
    dcanet_amd.models.augment.*      <- models/augment/{cva,semantic_level,SelfAttention_bn}.py
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["ops", "_lib"]
