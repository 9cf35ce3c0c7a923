"""tiatoolbox_amd: the per-patch inference hot path of tiatoolbox on MI355X (see README.md / DESIGN.md).

Sub-packages mirror the reference's layout (``tools``, ``utils``, ``models``, ``wsicore``) so that replacing
``tiatoolbox`` by ``tiatoolbox_amd`` in the imports of a pipeline switches the covered path over.  They are imported
lazily (PEP 562): ``import tiatoolbox_amd`` alone loads neither torch nor the HIP library.
"""

from __future__ import annotations

import importlib
import logging

__version__ = "0.1.0"

logger = logging.getLogger("tiatoolbox_amd")

_SUBMODULES = ("models", "tools", "utils", "wsicore", "distributed")

__all__ = ["__version__", "logger", *_SUBMODULES]


def __getattr__(name: str):
    if name in _SUBMODULES:
        return importlib.import_module(f"{__name__}.{name}")
    msg = f"module {__name__!r} has no attribute {name!r}"
    raise AttributeError(msg)
