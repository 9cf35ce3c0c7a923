"""Patch classification engine (API of reference ``tiatoolbox/models/engine/patch_predictor.py``)."""

from __future__ import annotations

import torch

from tiatoolbox_amd.models.engine.engine_abc import EngineABC
from tiatoolbox_amd.utils.misc import cast_to_min_dtype


class PatchPredictor(EngineABC):
    """Patch-level prediction: ``probabilities`` (optional) + ``predictions`` (ref. :88-679).

    Extra run-time kwargs on MI355X: ``compute_dtype`` ("float32" | "float16" | "bfloat16") and
    ``stain_normalizer`` (a fitted :class:`~tiatoolbox_amd.tools.stainnorm.StainNormalizer`; shorthand
    for ``model.preproc_func = StainNormPreproc(normalizer, default_preproc)``).
    """

    def __init__(self, model, batch_size: int = 8, num_workers: int = 0, weights=None, *,
                 device: str = "cpu", verbose: bool = True) -> None:
        super().__init__(model=model, batch_size=batch_size, num_workers=num_workers, weights=weights,
                         device=device, verbose=verbose)
        self.return_probabilities = False
        self.stain_normalizer = None
        self._default_preproc = self._get_model_attr("preproc_func")

    def _update_run_params(self, images, **kwargs):
        """ref. :448-549: ``probabilities`` are dropped unless THIS call passes ``return_probabilities=True`` (ref. :535-537:
        ``kwargs.get("return_probabilities")`` -- per call, not sticky)."""
        out = super()._update_run_params(images, **kwargs)
        if not self.return_probabilities:
            self.drop_keys.append("probabilities")
        elif "probabilities" in self.drop_keys:
            self.drop_keys = [k for k in self.drop_keys if k != "probabilities"]
        from tiatoolbox_amd.models.dataset.classification import StainNormPreproc

        model = self.model.module if hasattr(self.model, "module") else self.model
        if self.stain_normalizer is not None:
            model.preproc_func = StainNormPreproc(self.stain_normalizer, self._default_preproc)
        elif isinstance(model.preproc_func, StainNormPreproc) and getattr(self, "_installed_norm", False):
            model.preproc_func = self._default_preproc  # a later run(stain_normalizer=None) undoes the shorthand
        self._installed_norm = self.stain_normalizer is not None
        return out

    def post_process_patches(self, raw_predictions: dict, **_) -> dict:
        """``predictions = cast_to_min_dtype(argmax(probabilities))`` (ref. :321-380)."""
        postproc_func = self._get_model_attr("postproc_func")
        probs = raw_predictions["probabilities"]
        predictions = postproc_func(probs)
        if isinstance(predictions, torch.Tensor) and predictions.numel() == 0:
            raw_predictions["predictions"] = predictions.to(torch.uint8)
        else:
            raw_predictions["predictions"] = cast_to_min_dtype(predictions)
        return raw_predictions
