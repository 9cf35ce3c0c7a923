"""pytest configuration: ``gpu`` marker + shared seeded inputs."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def he_patches():
    from tiatoolbox_amd.utils import synth

    return synth.g_he(8, 256, 256, seed=1)


@pytest.fixture(scope="session")
def uniform_patches():
    from tiatoolbox_amd.utils import synth

    return synth.g_uniform(4, 64, 64, seed=0)


@pytest.fixture(scope="session")
def target_image():
    """The reference's packaged stain-norm target, committed as a 256x256 crop fixture."""
    p = ROOT / "tests" / "golden" / "target_crop_256.npy"
    return np.load(p)


@pytest.fixture(params=["auto", "direct"])
def conv_algo(request):
    """Both forms of the float32 3x3 / stride-1 block convolutions: the engines' default ``"auto"`` (Winograd F(2x2, 3x3) where the
    per-layer error-bound test covers the layer) and the audit mode ``"direct"``; GPU parity tests take this fixture and pass it on
    as the ``conv_algo`` run kwarg."""
    return request.param
