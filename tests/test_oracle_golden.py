"""Pin the CPU oracle (``oracle/``) to the reference's own golden vectors / real outputs.

* offline known answers copied *as data* from the reference's tests
  (``tests/test_utils.py:882-911``, ``tests/test_stainnorm.py:16-68``);
* ``tests/golden/stain_golden.npz``: outputs of the REAL reference modules executed in the
  build container by ``tests/golden/make_golden.py`` (absent cv2/skimage primitives shimmed).
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest

from oracle import stain
from tiatoolbox_amd.utils import synth

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "stain_golden.npz")


def test_contrast_enhancer_reference_golden():
    inp = np.array([[[37, 244, 193], [106, 235, 128], [71, 140, 47]],
                    [[103, 184, 72], [20, 188, 238], [126, 7, 0]],
                    [[137, 195, 204], [32, 203, 170], [101, 77, 133]]], dtype=np.uint8)
    exp = np.array([[[35, 255, 203], [110, 248, 133], [72, 146, 46]],
                    [[106, 193, 73], [17, 198, 251], [131, 3, 0]],
                    [[143, 205, 215], [30, 214, 178], [104, 78, 139]]], dtype=np.uint8)
    with pytest.raises(AssertionError):
        stain.contrast_enhancer(np.float32(inp), low_p=2, high_p=98)
    assert np.array_equal(stain.contrast_enhancer(inp, low_p=2, high_p=98), exp)


def test_helper_truth_tables():
    e = stain.vectors_in_correct_direction(np.ones([2, 2]))
    assert np.all(e == 1)
    e = np.ones([2, 2]); e[0, 0] = -1
    e = stain.vectors_in_correct_direction(e)
    assert np.all(e[:, 1] == 1) and e[0, 0] == 1 and e[1, 0] == -1
    e = np.ones([2, 2]); e[0, 1] = -1
    e = stain.vectors_in_correct_direction(e)
    assert np.all(e[:, 0] == 1) and e[0, 1] == 1 and e[1, 1] == -1
    v1, v2 = np.ones(3), np.zeros(3)
    assert np.all(stain.h_and_e_in_right_order(v1, v2) == np.array([v1, v2]))
    assert np.all(stain.h_and_e_in_right_order(v2, v1) == np.array([v1, v2]))
    d = np.zeros([20, 15])
    assert np.all(stain.dl_output_for_h_and_e(d) == d)
    d[1, :] = 1
    d2 = stain.dl_output_for_h_and_e(d)
    assert d2.shape == (2, 15) and np.all(d2 == d[[1, 0], :])
    with pytest.raises(ValueError, match=r"Stain matrix must have shape \(2, 3\) or \(3, 3\)."):
        stain.CustomExtractor(np.array([0.65, 0.70, 0.29]))


def test_empty_mask_error():
    with pytest.raises(ValueError, match="Empty tissue mask"):
        stain.get_luminosity_tissue_mask(np.zeros((100, 100, 3)), threshold=0)


@pytest.mark.parametrize("method", ["macenko", "ruifrok", "custom"])
def test_normalizers_match_real_reference(gold, target_image, method):
    sm = np.array([[0.60, 0.72, 0.34], [0.10, 0.95, 0.29]]) if method == "custom" else None
    norm = stain.get_normalizer(method, stain_matrix=sm)
    norm.fit(target_image.copy())
    np.testing.assert_allclose(norm.stain_matrix_target, gold[f"{method}_stain_matrix_target"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(norm.maxC_target, gold[f"{method}_maxC_target"], rtol=0, atol=1e-12)
    assert np.array_equal(norm.stain_matrix_target_RGB, gold[f"{method}_stain_matrix_target_RGB"])
    np.testing.assert_allclose(norm.target_concentrations[:64], gold[f"{method}_target_conc_head"], atol=1e-12)
    he = synth.g_he(3, 96, 96, seed=int(gold["he_seed"]))
    for imgs, key in ((gold["real_crops"], f"{method}_real"), (he, f"{method}_he")):
        out = np.stack([norm.transform(c.copy()) for c in imgs])
        assert np.array_equal(out, gold[key])


def test_mask_and_enhancer_match_real_reference(gold):
    crops = gold["real_crops"]
    assert np.array_equal(np.stack([stain.contrast_enhancer(c.copy()) for c in crops]), gold["ce_real"])
    assert np.array_equal(np.stack([stain.get_luminosity_tissue_mask(c.copy(), 0.8) for c in crops]),
                          gold["mask08_real"])
    he = synth.g_he(3, 96, 96, seed=int(gold["he_seed"]))
    assert np.array_equal(np.stack([stain.get_luminosity_tissue_mask(c.copy(), 0.85) for c in he]),
                          gold["mask085_he"])


def test_augment_matches_real_reference(gold):
    crops = gold["real_crops"]
    for k, ab in enumerate(gold["augment_ab"]):
        sm = stain.MacenkoExtractor().get_stain_matrix(crops[k].copy())
        out = stain.stain_augment(crops[k], sm, ab[:2], ab[2:], threshold=0.85, augment_background=bool(k))
        assert np.array_equal(out, gold["augment_real"][k])


def test_lab_known_colours():
    """Widely published OpenCV 8-bit Lab values for the primaries."""
    from oracle import cvref

    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], dtype=np.uint8)
    exp = np.array([[[255, 128, 128], [0, 128, 128], [136, 208, 195], [224, 42, 211], [82, 207, 20]]])
    assert np.array_equal(cvref.rgb2lab_u8(px), exp)
