"""Factory / loader contracts of the stain front-end, offline and GPU-free.

Ports of the reference's own tests: ``tests/test_stainnorm.py:101-119`` (``get_normalizer`` argument errors) and
``tests/test_utils.py:914-936`` (``load_stain_matrix``: .csv / .npy round trips, ``FileNotSupportedError``, ``TypeError``),
plus the unknown-method error of ``tools/stainnorm.py:396-403`` and the ``CustomExtractor`` shape check
(``tools/stainextract.py:86-91``).  None of these paths touches the device.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest

from tiatoolbox_amd.tools.stainnorm import get_normalizer
from tiatoolbox_amd.utils import misc
from tiatoolbox_amd.utils.exceptions import FileNotSupportedError, MethodNotSupportedError

RUIFROK = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]])


def test_get_normalizer_assertion():
    with pytest.raises(ValueError, match=r"`stain_matrix` is only defined when using `method_name`=\"custom\"."):
        get_normalizer("ruifrok", RUIFROK)
    for name in ("reinhard", "macenko", "vahadane", "Macenko", "REINHARD"):   # the check is case-insensitive (:396)
        with pytest.raises(ValueError, match="only defined when"):
            get_normalizer(name, RUIFROK)


def test_get_custom_normalizer_assertion():
    with pytest.raises(ValueError, match=r"`stain_matrix` is None when using `method_name`=\"custom\"."):
        get_normalizer("custom", None)
    with pytest.raises(ValueError, match="is None when using"):
        get_normalizer("Custom")


def test_unknown_method_and_order_of_checks():
    with pytest.raises(MethodNotSupportedError):
        get_normalizer("histogram-matching")
    with pytest.raises(MethodNotSupportedError):      # the name is checked before the stain matrix (:396-407)
        get_normalizer("histogram-matching", RUIFROK)


def test_factory_returns_the_reference_classes():
    from tiatoolbox_amd.tools import stainnorm
    from tiatoolbox_amd.tools.reinhard import ReinhardNormalizer

    assert isinstance(get_normalizer("Reinhard"), ReinhardNormalizer)
    assert isinstance(get_normalizer("ruifrok"), stainnorm.RuifrokNormalizer)
    assert isinstance(get_normalizer("MACENKO"), stainnorm.MacenkoNormalizer)
    assert isinstance(get_normalizer("vahadane"), stainnorm.VahadaneNormalizer)
    custom = get_normalizer("custom", RUIFROK)
    assert isinstance(custom, stainnorm.CustomNormalizer)
    assert np.array_equal(custom.extractor.stain_matrix, RUIFROK)
    with pytest.raises(ValueError, match="Stain matrix must have shape"):     # stainextract.py:86-91
        get_normalizer("custom", np.zeros((3, 2)))


def test_load_stain_matrix(tmp_path: Path):
    import pandas as pd

    with pytest.raises(FileNotSupportedError):
        misc.load_stain_matrix("/samplefile.xlsx")
    with pytest.raises(TypeError):
        misc.load_stain_matrix([1, 2, 3])
    pd.DataFrame(RUIFROK).to_csv(tmp_path / "sm.csv", index=False)
    assert np.all(misc.load_stain_matrix(tmp_path / "sm.csv") == RUIFROK)
    np.save(str(tmp_path / "sm.npy"), RUIFROK)
    assert np.all(misc.load_stain_matrix(tmp_path / "sm.npy") == RUIFROK)
    assert np.all(misc.load_stain_matrix(str(tmp_path / "sm.npy")) == RUIFROK)
    assert misc.load_stain_matrix(RUIFROK) is RUIFROK
    # the factory goes through the loader: a path is as good as an array
    assert np.all(get_normalizer("custom", tmp_path / "sm.csv").extractor.stain_matrix == RUIFROK)
    with pytest.raises(FileNotSupportedError):
        get_normalizer("custom", "/samplefile.xlsx")


def test_stain_extract_helpers_of_the_product():
    """The package's own host helpers (not the oracle's): ports of reference ``tests/test_stainnorm.py:16-68`` --
    ``CustomExtractor`` shape check, ``vectors_in_correct_direction``, ``h_and_e_in_right_order``, ``dl_output_for_h_and_e``."""
    from tiatoolbox_amd.tools import stainextract

    with pytest.raises(ValueError, match=r"Stain matrix must have shape \(2, 3\) or \(3, 3\)."):
        stainextract.CustomExtractor(np.array([0.65, 0.70, 0.29]))

    e_vect = stainextract.vectors_in_correct_direction(e_vectors=np.ones([2, 2]))
    assert np.all(e_vect == 1)
    e_vect = np.ones([2, 2])
    e_vect[0, 0] = -1
    e_vect = stainextract.vectors_in_correct_direction(e_vectors=e_vect)
    assert np.all(e_vect[:, 1] == 1) and e_vect[0, 0] == 1 and e_vect[1, 0] == -1
    e_vect = np.ones([2, 2])
    e_vect[0, 1] = -1
    e_vect = stainextract.vectors_in_correct_direction(e_vectors=e_vect)
    assert np.all(e_vect[:, 0] == 1) and e_vect[0, 1] == 1 and e_vect[1, 1] == -1

    v1, v2 = np.ones(3), np.zeros(3)
    assert np.all(stainextract.h_and_e_in_right_order(v1, v2) == np.array([v1, v2]))
    assert np.all(stainextract.h_and_e_in_right_order(v1=v2, v2=v1) == np.array([v1, v2]))

    dictionary = np.zeros([20, 15])
    assert np.all(stainextract.dl_output_for_h_and_e(dictionary=dictionary) == dictionary)
    dictionary[1, :] = 1
    ordered = stainextract.dl_output_for_h_and_e(dictionary=dictionary)
    assert ordered.shape == (2, 15) and np.all(ordered == dictionary[[1, 0], :])
