"""Deep feature extraction engine (API of reference ``tiatoolbox/models/engine/deep_feature_extractor.py``).

Same inference path as :class:`PatchPredictor` (device-resident batches, BN-folded trunk, the hand-written MFMA
convolutions for float32 BasicBlock trunks); what differs is the tail: the model's output (``CNNBackbone``: the pooled
feature vector) is returned untouched under ``"probabilities"`` -- no arg-max, no ``"predictions"`` key -- together with
``"coordinates"`` in WSI mode (ref. :70-141 constructor, :142-260 ``infer_wsi``, :262-290 ``post_process_patches``).
The reference spills feature chunks to zarr when host memory runs short; here features of a run stay in one device
tensor until they are copied out (a 100k-patch slide of 2048-d float32 features is 0.8 GB of 288 GB).
"""

from __future__ import annotations

from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor


class DeepFeatureExtractor(PatchPredictor):
    """Patch / WSI feature extractor: ``run()`` returns ``{"probabilities": features[, "coordinates"]}``."""

    def __init__(self, model, batch_size: int = 8, num_workers: int = 0, weights=None, *,
                 device: str = "cpu", verbose: bool = True) -> None:
        super().__init__(model=model, batch_size=batch_size, num_workers=num_workers, weights=weights,
                         device=device, verbose=verbose)
        self.process_prediction_per_batch = False
        self.return_probabilities = True

    def _update_run_params(self, images, **kwargs):
        kwargs["return_probabilities"] = True  # the features ARE the output (ref. :262-290 ignores the flag)
        return super()._update_run_params(images, **kwargs)

    def post_process_patches(self, raw_predictions: dict, **_) -> dict:
        """Features pass through unchanged (ref. :262-290)."""
        return raw_predictions
