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
    assert lib.tia_abi_version() == 6


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


def test_host_only_dispatch_queries():
    """``tia_conv3x3_geometry`` / ``tia_stain_stats_path`` answer on the host (no launch): which block geometry a 3x3 / stride-1
    convolution gets and which statistics kernel a patch shape gets -- the dispatch the GPU tests and the bench rely on.
    Band geometry: strips of ``bw`` columns x ``br`` real rows of the stacked batch with ``bw * br <= 256``, LDS patch
    ``(br + zero rows straddled + 2) * row_pitch <= 1728`` units, ``bw * strips == w``; only for "same" padding, only above 0.88 busy."""
    import ctypes

    import pytest

    from tiatoolbox_amd import _lib, build

    if not build.LIB_PATH.exists():
        pytest.skip("library not built (run __graft_entry__.build())")
    lib = _lib.load()

    def geometry(h, w, ho, wo, pad):
        geom = (ctypes.c_int32 * 4)()
        return lib.tia_conv3x3_geometry(h, w, ho, wo, pad, pad, geom), list(geom)

    # resnet maps of 224^2 patches: 56 / 28 / 14 / 7 on bands of real rows; of 256^2 patches: fixed geometries
    assert geometry(56, 56, 56, 56, 1) == (4, [8, 32, 40, 7])
    assert geometry(28, 28, 28, 28, 1) == (4, [4, 64, 24, 7])
    assert geometry(14, 14, 14, 14, 1) == (4, [14, 18, 64, 1])
    assert geometry(7, 7, 7, 7, 1) == (4, [7, 36, 36, 1])  # 36 real rows + the five zero rows they can straddle + 2 halo rows
    assert geometry(3, 3, 3, 3, 1)[0] == 0
    for side in (64, 32, 16, 128, 512):
        assert geometry(side, side, side, side, 1) == (1, [0, 0, 0, 0])
    assert geometry(8, 8, 8, 8, 1)[0] == 2
    # the kernel a float32 convolution runs on (0 slice / 1 tap reuse / 2 LDS-DMA ring): resnet18 on a 1024-patch batch (a host
    # without a GPU answers for 256 CUs) -- stride-2 3x3 on the gathering ring while its rounds are >= 85 % full, small
    # batches (and a 7 x 7 launch of 1.56 workgroup rounds) on the slice kernel, 1x1 down-sampling on the ring
    def route(n, hw, cin, cout, k, stride, pad):
        ho = (hw + 2 * pad - k) // stride + 1
        return lib.tia_conv2d_route_f32(n, hw, hw, cin, cout, k, k, stride, pad, pad, ho, ho)

    assert route(1024, 64, 64, 64, 3, 1, 1) == 1 and route(1024, 8, 512, 512, 3, 1, 1) == 1
    # bands of 7 x 7 maps: taken for one round of workgroups or for well-filled launches, not for 1.56 rounds (1024 patches)
    assert route(4096, 7, 512, 512, 3, 1, 1) == 1 and route(256, 7, 512, 512, 3, 1, 1) == 1 and route(1024, 14, 256, 256, 3, 1, 1) == 1
    assert route(1024, 64, 64, 128, 3, 2, 1) == 2 and route(1024, 32, 128, 256, 3, 2, 1) == 2 and route(1024, 16, 256, 512, 3, 2, 1) == 2  # noqa: PLR2004
    assert route(1024, 56, 64, 128, 3, 2, 1) == 2 and route(1024, 28, 128, 256, 3, 2, 1) == 0 and route(1024, 7, 512, 512, 3, 1, 1) == 0  # noqa: PLR2004
    assert route(1024, 64, 64, 128, 1, 2, 0) == 2 and route(4, 64, 64, 128, 3, 2, 1) == 0 and route(1024, 64, 64, 64, 1, 1, 0) == 0  # noqa: PLR2004
    assert route(0, 64, 64, 128, 3, 2, 1) < 0
    # the query runs the entry point's own shape checks: what tia_conv2d_nhwc_f32 rejects (cin % 32, cout % 64) is rejected here
    # with the same code (TIA_ESIZE = -3), and a batch beyond 2 GiB of input is answered for one full group
    assert route(8, 16, 48, 64, 3, 1, 1) == -3 and route(8, 16, 16, 64, 3, 1, 1) == -3 and route(8, 16, 64, 96, 3, 1, 1) == -3  # noqa: PLR2004
    assert route(100000, 64, 64, 64, 3, 1, 1) == 1
    # valid convolutions (HoVer-Net's decoder): bands of real output rows as well
    assert geometry(164, 164, 162, 162, 0) == (4, [27, 9, 116, 6]) and geometry(64, 64, 62, 62, 0)[0] == 1  # valid: bands too, for a clear gain
    assert geometry(48, 48, 47, 47, 0)[0] == 1 and geometry(50, 50, 49, 49, 0)[0] == 0  # (asymmetric borders: fixed geometries or none)
    for h in range(9, 130):
        kind, (bw, br, pitch, strips) = geometry(h, h, h, h, 1)
        if kind == 3:  # noqa: PLR2004  (zero rows among the GEMM rows: where the larger patch of kind 4 would cost a band row)
            assert bw * strips == h and bw * br <= 256 and (br + 2) * pitch <= 1728 and pitch >= 4 * (bw + 2)  # noqa: PLR2004
            assert bw * br / 256 * h / (h + 1) >= 0.88  # noqa: PLR2004
        if kind == 4:  # noqa: PLR2004
            assert bw * strips == h and bw * br <= 256 and pitch >= 4 * (bw + 2)
            assert (br + (br + h - 2) // h + 2) * pitch <= 1728 and bw * br / 256 >= 0.88  # noqa: PLR2004
    # rectangular maps: the strip width divides the map width
    kind, (bw, br, pitch, strips) = geometry(40, 56, 40, 56, 1)
    assert kind == 4 and bw * strips == 56  # noqa: PLR2004

    from tiatoolbox_amd.tools import _stain_device as dev

    for mode, expect in ((_lib.MODE_MACENKO, 1), (_lib.MODE_FIXED, 1), (_lib.MODE_VAHADANE, 0)):
        kw = {"stain_fixed": [[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]]} if mode == _lib.MODE_FIXED else {}
        import numpy as np

        prm = dev.make_params(mode=mode, **({k: np.array(v) for k, v in kw.items()}))
        assert lib.tia_stain_stats_path(256, 256, ctypes.byref(prm)) == expect
        assert lib.tia_stain_stats_path(224, 224, ctypes.byref(prm)) == expect
        assert lib.tia_stain_stats_path(300, 300, ctypes.byref(prm)) == 0     # does not fit 16 groups per thread
        assert lib.tia_stain_stats_path(37, 41, ctypes.byref(prm)) == 0       # 1517 pixels: no whole 4-pixel groups
        assert lib.tia_stain_stats_path(32, 32, ctypes.byref(prm)) == 0       # fewer pixels than the window-placing sample
    for select_mode in (1, 2):
        prm = dev.make_params(mode=_lib.MODE_MACENKO, select_mode=select_mode)
        assert lib.tia_stain_stats_path(256, 256, ctypes.byref(prm)) == 0     # the audit modes keep the streaming kernel
    assert lib.tia_stain_stats_path(0, 5, ctypes.byref(prm)) == -1            # TIA_EINVAL
    assert lib.tia_clear_last_error() in (0, 3, 100, 101)                      # callable without a device (no device: hipErrorNoDevice)
