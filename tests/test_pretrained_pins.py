"""Pins against the REAL pretrained checkpoints -- run when they are supplied, skipped (with the reason) otherwise.

No network here or on the GPU box, so neither the weights (HuggingFace ``TIACentre/TIAToolbox_pretrained_weights``) nor
the reference's sample images are reachable; every model-parity test in this repo therefore uses seeded random weights.
These tests are the reference's own checkpoint-level known answers, armed by pointing ``TIA_WEIGHTS_DIR`` at a directory with

* ``resnet18-kather100k.pth``, ``kather_patch1.tif``, ``kather_patch2.tif``
  (``/root/reference/tests/engines/test_patch_predictor.py:263-290``: classes [6, 3], max probabilities 1.0 / 0.9999911785);
* ``hovernet_fast-pannuke.pth`` and ``wsi4_1k_1k_patches.npy`` -- the three 256 x 256 uint8 patches that test reads from
  ``wsi4_1k_1k.svs`` at 0.5 mpp, locations (0, 0) and (512, 512), plus the all-zero patch, stacked ``[3, 256, 256, 3]``
  (``/root/reference/tests/engines/test_nucleus_instance_segmentor.py:44-82``: 62 / 33 / 0 nuclei); the patches are
  pre-extracted because this repo has no SVS reader (out of scope).

``local_pretrained_weights`` (the product's look-up of the reference's download cache) is tested offline.
"""

from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import pytest

WEIGHTS = Path(os.environ["TIA_WEIGHTS_DIR"]) if os.environ.get("TIA_WEIGHTS_DIR") else None


def _need(*names: str) -> Path:
    if WEIGHTS is None:
        pytest.skip("TIA_WEIGHTS_DIR is not set (pretrained checkpoints are not reachable from here)")
    missing = [n for n in names if not (WEIGHTS / n).is_file()]
    if missing:
        pytest.skip(f"TIA_WEIGHTS_DIR lacks {missing}")
    return WEIGHTS


def test_local_weights_lookup(tmp_path, monkeypatch):
    import torch

    from tiatoolbox_amd.models.architecture import get_pretrained_model, local_pretrained_weights

    monkeypatch.delenv("TIA_WEIGHTS_DIR", raising=False)
    monkeypatch.setenv("TIATOOLBOX_HOME", str(tmp_path / "home"))
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    assert local_pretrained_weights("resnet18-kather100k") is None
    model, _ = get_pretrained_model("resnet18-kather100k", seed=3)
    sd = {k: torch.full_like(v, 0.25) if v.is_floating_point() else v for k, v in model.state_dict().items()}
    (tmp_path / "home" / "models").mkdir(parents=True)
    torch.save(sd, tmp_path / "home" / "models" / "resnet18-kather100k.pth")
    assert local_pretrained_weights("resnet18-kather100k") == tmp_path / "home" / "models" / "resnet18-kather100k.pth"
    loaded, cfg = get_pretrained_model("resnet18-kather100k")
    assert float(loaded.classifier.weight.mean()) == 0.25 and list(cfg.patch_input_shape) == [224, 224]
    # TIA_WEIGHTS_DIR wins over the cache
    (tmp_path / "w").mkdir()
    torch.save({k: torch.full_like(v, 0.5) if v.is_floating_point() else v for k, v in sd.items()},
               tmp_path / "w" / "resnet18-kather100k.pth")
    monkeypatch.setenv("TIA_WEIGHTS_DIR", str(tmp_path / "w"))
    assert float(get_pretrained_model("resnet18-kather100k")[0].classifier.weight.mean()) == 0.5
    # an explicit path wins over both
    explicit = get_pretrained_model("resnet18-kather100k", tmp_path / "home" / "models" / "resnet18-kather100k.pth")[0]
    assert float(explicit.classifier.weight.mean()) == 0.25


@pytest.mark.gpu
def test_patch_predictor_kather100k_output():
    """Reference ``test_patch_predictor_kather100k_output`` for resnet18: classes [6, 3], max probability per patch 1.0 and
    0.9999911785125732 (reference tolerance: ``abs < 1e-3``-scale equality of the float32 softmax; asserted here at 1e-5)."""
    root = _need("resnet18-kather100k.pth", "kather_patch1.tif", "kather_patch2.tif")
    from PIL import Image

    from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor

    patches = np.stack([np.asarray(Image.open(root / n).convert("RGB")) for n in ("kather_patch1.tif", "kather_patch2.tif")])
    eng = PatchPredictor(model="resnet18-kather100k", batch_size=32, device="cuda")
    out = eng.run(patches, patch_mode=True, return_probabilities=True)
    probs = np.asarray(out["probabilities"])
    assert list(np.asarray(out["predictions"])) == [6, 3]
    np.testing.assert_allclose(probs.max(axis=1), [1.0, 0.9999911785125732], atol=1e-5)


@pytest.mark.gpu
def test_hovernet_fast_pannuke_nuclei_counts():
    """Reference ``test_mtsegmentor_patches``: 62 / 33 / 0 nuclei on the three sample patches."""
    root = _need("hovernet_fast-pannuke.pth", "wsi4_1k_1k_patches.npy")
    from tiatoolbox_amd.models.engine.nucleus_instance_segmentor import NucleusInstanceSegmentor

    patches = np.load(root / "wsi4_1k_1k_patches.npy")
    assert patches.shape == (3, 256, 256, 3) and patches.dtype == np.uint8
    seg = NucleusInstanceSegmentor(model="hovernet_fast-pannuke", batch_size=32, device="cuda")
    out = seg.run(images=patches, patch_mode=True, return_probabilities=True)
    counts = [len(np.asarray(b)) for b in out["box"]]
    assert counts == [62, 33, 0], counts
    for field in ("centroid", "contours", "prob", "type"):
        assert [len(np.asarray(v)) for v in out[field]] == counts
