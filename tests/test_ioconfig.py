"""Known answers of the reference's offline ``tests/engines/test_ioconfig.py`` (validation errors, ``scale_to_highest``
for mpp / baseline / power / unknown units, ``to_baseline`` with and without ``save_resolution``) asserted on this
repo's ``io_config`` classes, plus the docstring examples of reference ``io_config.py:42-108,263-323,418-461``."""

from __future__ import annotations

import numpy as np
import pytest

from tiatoolbox_amd.models import IOSegmentorConfig, ModelIOConfigABC
from tiatoolbox_amd.models.engine.io_config import IOInstanceSegmentorConfig, IOPatchPredictorConfig


def test_validation_error_io_config():
    with pytest.raises(ValueError, match=r".*Multiple resolution units found.*"):
        ModelIOConfigABC(input_resolutions=[{"units": "baseline", "resolution": 1.0}, {"units": "mpp", "resolution": 0.25}],
                         patch_input_shape=(224, 224))
    with pytest.raises(ValueError, match=r"Invalid resolution units.*"):
        ModelIOConfigABC(input_resolutions=[{"units": "level", "resolution": 1.0}], patch_input_shape=(224, 224))


def test_scale_to_highest_mpp():
    res = [{"units": "mpp", "resolution": 0.25}, {"units": "mpp", "resolution": 0.5}]
    np.testing.assert_allclose(ModelIOConfigABC.scale_to_highest(res, units="mpp"), np.array([1.0, 0.5]))
    np.testing.assert_allclose(ModelIOConfigABC.scale_to_highest(res[::-1], units="mpp"), np.array([0.5, 1.0]))


def test_scale_to_highest_baseline_power_and_unknown_units():
    res = [{"units": "baseline", "resolution": 2.0}, {"units": "baseline", "resolution": 4.0}]
    assert ModelIOConfigABC.scale_to_highest(res, units="baseline") == [2.0, 4.0]
    res = [{"units": "power", "resolution": 10}, {"units": "power", "resolution": 5}]
    np.testing.assert_allclose(ModelIOConfigABC.scale_to_highest(res, units="power"), np.array([1.0, 0.5]))
    with pytest.raises(ValueError, match="Unknown units"):
        ModelIOConfigABC.scale_to_highest([{"units": "mpp", "resolution": 1.0}], units="unknown")


def test_to_baseline_without_save_resolution():
    cfg = ModelIOConfigABC(input_resolutions=[{"units": "mpp", "resolution": 0.5}],
                           output_resolutions=[{"units": "mpp", "resolution": 1.0}], patch_input_shape=(224, 224),
                           stride_shape=(224, 224))
    new_cfg = cfg.to_baseline()
    assert not hasattr(new_cfg, "save_resolution") or new_cfg.save_resolution is None
    assert new_cfg.input_resolutions == [{"units": "baseline", "resolution": 1.0}]
    assert new_cfg.output_resolutions == [{"units": "baseline", "resolution": 0.5}]
    seg = IOSegmentorConfig(input_resolutions=[{"units": "mpp", "resolution": 0.5}],
                            output_resolutions=[{"units": "mpp", "resolution": 1.0}], patch_input_shape=(224, 224),
                            patch_output_shape=(112, 112), stride_shape=(224, 224), save_resolution=None)
    assert seg.to_baseline().save_resolution is None


def test_to_baseline_with_save_resolution_and_instance_fields():
    """Docstring examples of the reference: every resolution is rescaled against the highest one in the config."""
    seg = IOSegmentorConfig(input_resolutions=[{"units": "mpp", "resolution": 0.25}, {"units": "mpp", "resolution": 0.5}],
                            output_resolutions=[{"units": "mpp", "resolution": 0.5}], patch_input_shape=[2048, 2048],
                            patch_output_shape=[1024, 1024], stride_shape=[512, 512],
                            save_resolution={"units": "mpp", "resolution": 1.0})
    base = seg.to_baseline()
    assert [v["resolution"] for v in base.input_resolutions] == [1.0, 0.5]
    assert base.output_resolutions == [{"units": "baseline", "resolution": 0.5}]
    assert base.save_resolution == {"units": "baseline", "resolution": 0.25}
    assert seg.highest_input_resolution == {"units": "mpp", "resolution": 0.25} and seg.resolution_unit == "mpp"
    inst = IOInstanceSegmentorConfig(input_resolutions=[{"units": "mpp", "resolution": 0.25}],
                                     output_resolutions=[{"units": "mpp", "resolution": 0.25}] * 3, margin=128,
                                     tile_shape=[1024, 1024], patch_input_shape=[256, 256], patch_output_shape=[164, 164],
                                     stride_shape=[164, 164], save_resolution={"units": "mpp", "resolution": 0.25})
    b = inst.to_baseline()
    assert b.margin == 128 and list(b.tile_shape) == [1024, 1024] and len(b.output_resolutions) == 3
    assert all(v == {"units": "baseline", "resolution": 1.0} for v in b.output_resolutions)
    power = IOPatchPredictorConfig(input_resolutions=[{"units": "power", "resolution": 20}, {"units": "power", "resolution": 40}],
                                   patch_input_shape=(224, 224))
    assert power.highest_input_resolution["resolution"] == 40 and tuple(power.stride_shape) == (224, 224)
    assert [v["resolution"] for v in power.to_baseline().input_resolutions] == [0.5, 1.0]
