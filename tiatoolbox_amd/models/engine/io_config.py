"""Model I/O configuration dataclasses (API of reference ``tiatoolbox/models/engine/io_config.py``)."""

from __future__ import annotations

from dataclasses import dataclass, field, replace

import numpy as np

_UNITS = ("power", "baseline", "mpp")


@dataclass
class ModelIOConfigABC:
    """Patch shape / stride / resolutions of a model (ref. :14-212)."""

    input_resolutions: list[dict]
    patch_input_shape: list[int] | np.ndarray | tuple[int, int]
    stride_shape: list[int] | np.ndarray | tuple[int, int] = None
    output_resolutions: list[dict] = field(default_factory=list)
    ignore_index: int | None = (None,)

    def __post_init__(self) -> None:
        if self.stride_shape is None:
            self.stride_shape = self.patch_input_shape
        self.resolution_unit = self.input_resolutions[0]["units"]
        pick = min if self.resolution_unit == "mpp" else max
        self.highest_input_resolution = pick(self.input_resolutions, key=lambda x: x["resolution"])
        self._validate()

    def _validate(self) -> None:
        units = {v["units"] for v in self.input_resolutions + self.output_resolutions}
        if len(units) != 1:
            msg = f"Multiple resolution units found: `{units}`. Mixing resolution units is not allowed."
            raise ValueError(msg)
        if units.pop() not in _UNITS:
            msg = f"Invalid resolution units `{units}`."
            raise ValueError(msg)

    @staticmethod
    def scale_to_highest(resolutions: list[dict], units: str):
        """Scale factors w.r.t. the highest resolution in the list (ref. :111-175)."""
        vals = [v["resolution"] for v in resolutions]
        if units not in _UNITS:
            msg = f"Unknown units `{units}`. Units should be one of 'baseline', 'mpp' or 'power'."
            raise ValueError(msg)
        if units == "baseline":
            return vals
        if units == "mpp":
            return np.min(vals) / np.array(vals)
        return np.array(vals) / np.max(vals)

    def _baseline_parts(self):
        resolutions = self.input_resolutions + self.output_resolutions
        save_resolution = getattr(self, "save_resolution", None)
        if save_resolution is not None:
            resolutions = [*resolutions, save_resolution]
        sf = self.scale_to_highest(resolutions, self.resolution_unit)
        n_in, n_out = len(self.input_resolutions), len(self.output_resolutions)
        ins = [{"units": "baseline", "resolution": v} for v in sf[:n_in]]
        outs = [{"units": "baseline", "resolution": v} for v in sf[n_in:n_in + n_out]]
        return sf, ins, outs

    def to_baseline(self):
        _, ins, outs = self._baseline_parts()
        return replace(self, input_resolutions=ins, output_resolutions=outs)


@dataclass
class IOSegmentorConfig(ModelIOConfigABC):
    """Segmentation I/O (ref. :215-323)."""

    patch_output_shape: list[int] | np.ndarray | tuple[int, int] = None
    save_resolution: dict = None

    def to_baseline(self):
        sf, ins, outs = self._baseline_parts()
        save = None
        if self.save_resolution is not None:
            save = {"units": "baseline", "resolution": sf[-1]}
        return replace(self, input_resolutions=ins, output_resolutions=outs, save_resolution=save)


@dataclass
class IOPatchPredictorConfig(ModelIOConfigABC):
    """Patch-classification I/O (ref. :326-366)."""


@dataclass
class IOInstanceSegmentorConfig(IOSegmentorConfig):
    """Instance-segmentation I/O with tile processing (ref. :369-461)."""

    margin: int = None
    tile_shape: tuple[int, int] = None

    def to_baseline(self):
        base = super().to_baseline()
        return replace(base, margin=self.margin, tile_shape=self.tile_shape)
