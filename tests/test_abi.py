"""The C-ABI shared library loads and exports every symbol ``include/*.h`` declares."""

from __future__ import annotations

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols() -> list[str]:
    names: list[str] = []
    for h in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names += re.findall(r"^\s*(?:int|void|size_t)\s+(tia_\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_entry_points():
    syms = _declared_symbols()
    assert "tia_stain_stats_u8" in syms and "tia_stain_apply_u8" in syms


def test_library_exports_every_declared_symbol():
    from tiatoolbox_amd import build

    path = build.LIB_PATH
    if not path.exists():
        pytest.skip("library not built (run __graft_entry__.build())")
    try:
        lib = ctypes.CDLL(str(path))
    except OSError as exc:  # pragma: no cover
        pytest.fail(f"cannot dlopen {path}: {exc}")
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"missing exports: {missing}"
    lib.tia_abi_version.restype = ctypes.c_int
    assert lib.tia_abi_version() == 4


def test_python_binding_covers_header():
    from tiatoolbox_amd import _lib

    assert sorted(_lib._SIGNATURES) == _declared_symbols()


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors == what a C compiler makes of ``include/tiatoolbox_amd.h`` (sizes and every field offset)."""
    import shutil
    import subprocess

    from tiatoolbox_amd import _lib

    assert ctypes.sizeof(_lib.StainTables) == 256 * 8 + 256 * 4 + 3 * 256 * 4
    assert ctypes.sizeof(_lib.StainParams) == 8 * (5 + 6 + 6 + 2) + 4 * 4 + 8 * 2 + 4 * 4
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    fields = [name for name, _ in _lib.StainParams._fields_]  # noqa: SLF001
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "tiatoolbox_amd.h"', "int main(void) {",
             '  printf("%zu %zu\\n", sizeof(tia_stain_params), sizeof(tia_stain_tables));']
    lines += [f'  printf("{f} %zu\\n", offsetof(tia_stain_params, {f}));' for f in fields]
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    sizes = [int(v) for v in out[0].split()]
    assert sizes == [ctypes.sizeof(_lib.StainParams), ctypes.sizeof(_lib.StainTables)]
    for line in out[1:]:
        if line.strip():
            name, off = line.split()
            assert getattr(_lib.StainParams, name).offset == int(off), name


def test_no_cpu_fallback_without_gpu():
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tiatoolbox_amd import _lib
    from tiatoolbox_amd.tools.stainnorm import get_normalizer

    norm = get_normalizer("macenko")
    with pytest.raises(_lib.HipLibraryError):
        norm.fit(np.zeros((8, 8, 3), np.uint8))


def test_package_level_names_of_the_reference_layout():
    """``s/tiatoolbox/tiatoolbox_amd/`` in a pipeline's imports must resolve for the covered path
    (reference ``tiatoolbox/__init__.py``, ``models/__init__.py``, ``tools/__init__.py``, ``utils/__init__.py``)."""
    import importlib
    import sys

    import tiatoolbox_amd

    assert tiatoolbox_amd.logger.name == "tiatoolbox_amd" and tiatoolbox_amd.__version__
    from tiatoolbox_amd import models, tools, utils
    from tiatoolbox_amd.models import (HoVerNet, IOInstanceSegmentorConfig, IOPatchPredictorConfig, IOSegmentorConfig,  # noqa: F401
                                       ModelIOConfigABC, MultiTaskSegmentor, NucleusInstanceSegmentor, PatchDataset,
                                       PatchPredictor, SemanticSegmentor)
    from tiatoolbox_amd.models.dataset import predefined_preproc_func  # noqa: F401
    from tiatoolbox_amd.models.engine.nucleus_instance_segmentor import NucleusInstanceSegmentor as Moved

    assert Moved is NucleusInstanceSegmentor
    assert models.engine.patch_predictor.PatchPredictor is PatchPredictor
    assert tools.stainnorm.get_normalizer and tools.tissuemask.OtsuTissueMasker and tools.stainaugment.StainAugmentor
    assert utils.misc.get_luminosity_tissue_mask and utils.transforms.rgb2od and utils.exceptions.MethodNotSupportedError
    for pkg in ("tiatoolbox_amd", "tiatoolbox_amd.models", "tiatoolbox_amd.tools", "tiatoolbox_amd.utils"):
        mod = sys.modules[pkg]
        for name in mod.__all__:
            assert getattr(mod, name) is not None, (pkg, name)
    import pytest

    assert importlib.import_module("tiatoolbox_amd.models").DeepFeatureExtractor.__name__ == "DeepFeatureExtractor"
    with pytest.raises(AttributeError):
        importlib.import_module("tiatoolbox_amd.models").NucleusDetector  # noqa: B018  (out of scope)
