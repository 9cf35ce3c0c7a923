"""Multi-head segmentation engines in patch mode (API of reference
``tiatoolbox/models/engine/multi_task_segmentor.py`` and ``nucleus_instance_segmentor.py``).

``infer_patches`` keeps every head's output resident on the GPU; ``post_process_patches`` runs the
model's *batched* device post-processing (HoVer-Net: Sobel/energy/CCL/watershed kernels over all
patches of a chunk at once) instead of a Python loop over patches.  Tile-mode WSI stitching
(reference :836-1554) is outside this round's scope.
"""

from __future__ import annotations

import warnings

import numpy as np
import torch

from tiatoolbox_amd.models.engine.engine_abc import EngineABC


class MultiTaskSegmentor(EngineABC):
    """Multi-task (here: nuclei instance) segmentation, patch mode (ref. :229-3829)."""

    def __init__(self, model, batch_size: int = 8, num_workers: int = 0, weights=None, *, device: str = "cpu",
                 verbose: bool = True) -> None:
        super().__init__(model=model, batch_size=batch_size, num_workers=num_workers, weights=weights, device=device,
                         verbose=verbose)
        self.return_probabilities = False
        self.fold_batchnorm = False  # HoVer-Net's pre-activation BN->ReLU->conv order cannot be folded forwards
        self.tasks = set(getattr(self.model, "tasks", []))

    def _update_run_params(self, images, **kwargs):
        if kwargs.get("return_labels"):
            msg = "`return_labels` is not supported for MultiTaskSegmentor."
            raise ValueError(msg)  # ref. :2149-2156
        self.return_probabilities = kwargs.get("return_probabilities", self.return_probabilities)
        return super()._update_run_params(images, **kwargs)

    def post_process_patches(self, raw_predictions: dict, **_) -> dict:
        """Per-patch ``postproc`` for every patch, batched on the device (ref. :733-834, :1556-1685)."""
        heads = raw_predictions["probabilities"]
        model = self.model.module if hasattr(self.model, "module") else self.model
        results: list[dict] = []
        n = heads[0].shape[0]
        on_gpu = heads[0].is_cuda
        chunk = 2048
        for s in range(0, n, chunk):
            part = [h[s:s + chunk] for h in heads]
            if on_gpu and hasattr(model, "postproc_batch"):
                results += model.postproc_batch(part[0], part[1], part[2] if len(part) > 2 else None)  # noqa: PLR2004
            else:
                postproc = self._get_model_attr("postproc_func")
                for i in range(part[0].shape[0]):
                    results.append(postproc([p[i].cpu().numpy() for p in part], offset=(0, 0))[0])
        out: dict = {}
        task = results[0]["task_type"] if results else "nuclei_segmentation"
        self.tasks = {task}
        out["predictions"] = (np.stack([r["predictions"] for r in results]) if results
                              else np.empty((0,), dtype=np.int32))
        for key in ("box", "centroid", "contours", "prob", "type"):
            out[key] = [r["info_dict"][key] for r in results]
        if self.return_probabilities:
            out["probabilities"] = [h.cpu().numpy() if isinstance(h, torch.Tensor) else h for h in heads]
        return out

    def save_predictions(self, processed_predictions: dict, output_type: str, **_):
        """Single task: the task dict is flattened into the top level, ``seg_type`` dropped (ref. :1695-1704)."""
        return {k: v for k, v in processed_predictions.items() if k not in self.drop_keys}


class NucleusInstanceSegmentor(MultiTaskSegmentor):
    """Deprecated alias kept by the reference (``nucleus_instance_segmentor.py:18-174``)."""

    def __init__(self, model, batch_size: int = 8, num_workers: int = 0, weights=None, *, device: str = "cpu",
                 verbose: bool = True) -> None:
        warnings.warn("NucleusInstanceSegmentor is deprecated and will be removed in a future release. "
                      "Use MultiTaskSegmentor instead.", DeprecationWarning, stacklevel=2)
        super().__init__(model=model, batch_size=batch_size, num_workers=num_workers, weights=weights, device=device,
                         verbose=verbose)
