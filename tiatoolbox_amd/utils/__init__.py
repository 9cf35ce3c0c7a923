"""Utilities (sub-modules of reference ``tiatoolbox/utils/__init__.py`` that are on the covered path)."""

from __future__ import annotations

import importlib

__all__ = ["exceptions", "misc", "transforms"]


def __getattr__(name: str):
    if name in (*__all__, "cvtables", "synth"):
        return importlib.import_module(f"{__name__}.{name}")
    msg = f"module {__name__!r} has no attribute {name!r}"
    raise AttributeError(msg)
