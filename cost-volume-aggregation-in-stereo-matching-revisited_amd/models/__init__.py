"""Model registry, as the reference's models/__init__.py:1-7 intends it (`__models__`)."""
from ._bootstrap import ensure as _ensure

_ensure()

from .gwcnet_dca_g import GwcNet, GwcNet_G, GwcNet_GC  # noqa: E402,F401

__models__ = {
    "gwcnet-g": GwcNet_G,
    "gwcnet-gc": GwcNet_GC,
}
