"""Model registry (API of reference ``tiatoolbox/models/architecture/__init__.py:70-178``).

Only the entries of the benchmark configs are registered.  Pretrained weights live on the
HuggingFace hub (``TIACentre/TIAToolbox_pretrained_weights``), unreachable from here: pass a
local ``.pth`` via ``weights=``, or keep ``<name>.pth`` where the reference caches its
downloads -- ``$TIA_WEIGHTS_DIR``, ``$TIATOOLBOX_HOME/models`` or ``~/.tiatoolbox/models``
(``fetch_pretrained_weights``, ref. :27-68); otherwise the model keeps its seeded random
initialisation and a warning is logged.  Parameter names match the reference so its files
load unchanged.
"""

from __future__ import annotations

import logging
from pathlib import Path

import torch

from tiatoolbox_amd.models.dataset.classification import predefined_preproc_func
from tiatoolbox_amd.models.engine.io_config import (
    IOInstanceSegmentorConfig,
    IOPatchPredictorConfig,
    IOSegmentorConfig,
)

logger = logging.getLogger("tiatoolbox_amd")

# mirrors tiatoolbox/data/pretrained_model.yaml (:15-28, :594-614, :642-672)
PRETRAINED_INFO = {
    "resnet18-kather100k": {
        "architecture": ("vanilla.CNNModel", {"backbone": "resnet18", "num_classes": 9}),
        "ioconfig": (IOPatchPredictorConfig, {
            "patch_input_shape": [224, 224], "stride_shape": [224, 224],
            "input_resolutions": [{"resolution": 0.5, "units": "mpp"}]}),
        "dataset": "kather100k",
    },
    "resnet34-kather100k": {
        "architecture": ("vanilla.CNNModel", {"backbone": "resnet34", "num_classes": 9}),
        "ioconfig": (IOPatchPredictorConfig, {
            "patch_input_shape": [224, 224], "stride_shape": [224, 224],
            "input_resolutions": [{"resolution": 0.5, "units": "mpp"}]}),
        "dataset": "kather100k",
    },
    "resnet50-kather100k": {
        "architecture": ("vanilla.CNNModel", {"backbone": "resnet50", "num_classes": 9}),
        "ioconfig": (IOPatchPredictorConfig, {
            "patch_input_shape": [224, 224], "stride_shape": [224, 224],
            "input_resolutions": [{"resolution": 0.5, "units": "mpp"}]}),
        "dataset": "kather100k",
    },
    "fcn_resnet50_unet-bcss": {
        "architecture": ("unet.UNetModel", {"num_input_channels": 3, "num_output_channels": 5, "encoder": "resnet50",
                                            "decoder_block": [3, 3]}),
        "ioconfig": (IOSegmentorConfig, {
            "input_resolutions": [{"units": "mpp", "resolution": 0.25}],
            "output_resolutions": [{"units": "mpp", "resolution": 0.25}],
            "patch_input_shape": [1024, 1024], "patch_output_shape": [512, 512], "stride_shape": [450, 450],
            "save_resolution": {"units": "mpp", "resolution": 0.25}, "ignore_index": 0}),
    },
    "hovernet_fast-pannuke": {
        "architecture": ("hovernet.HoVerNet", {
            "num_types": 6, "mode": "fast",
            "nuc_type_dict": {0: "Background", 1: "Neoplastic", 2: "Inflammatory", 3: "Connective", 4: "Dead",
                              5: "Non-Neoplastic Epithelial"}}),
        "ioconfig": (IOInstanceSegmentorConfig, {
            "input_resolutions": [{"units": "mpp", "resolution": 0.25}],
            "output_resolutions": [{"units": "mpp", "resolution": 0.25}] * 3,
            "margin": 128, "tile_shape": [1024, 1024], "patch_input_shape": [256, 256],
            "patch_output_shape": [164, 164], "stride_shape": [164, 164],
            "save_resolution": {"units": "mpp", "resolution": 0.25}, "ignore_index": 0}),
    },
    "hovernetplus-oed": {  # pretrained_model.yaml:758-793
        "architecture": ("hovernetplus.HoVerNetPlus", {
            "num_types": 3, "num_layers": 5,
            "nuc_type_dict": {0: "Background", 1: "Other", 2: "Epithelial"},
            "layer_type_dict": {0: "Background", 1: "Other Tissue", 2: "Basal Epithelium", 3: "(Core) Epithelium",
                                4: "Keratin"}}),
        "ioconfig": (IOInstanceSegmentorConfig, {
            "input_resolutions": [{"units": "mpp", "resolution": 0.50}],
            "output_resolutions": [{"units": "mpp", "resolution": 0.50}] * 4,
            "margin": 128, "tile_shape": [2048, 2048], "patch_input_shape": [256, 256],
            "patch_output_shape": [164, 164], "stride_shape": [164, 164],
            "save_resolution": {"units": "mpp", "resolution": 0.50}, "ignore_index": 0}),
    },
}


def local_pretrained_weights(model_name: str) -> Path | None:
    """``<model_name>.pth`` in the first existing of ``$TIA_WEIGHTS_DIR``, ``$TIATOOLBOX_HOME/models``,
    ``~/.tiatoolbox/models`` (the reference's download cache, ``architecture/__init__.py:57-68``); ``None`` if absent."""
    import os

    roots = []
    if os.environ.get("TIA_WEIGHTS_DIR"):
        roots.append(Path(os.environ["TIA_WEIGHTS_DIR"]))
    if os.environ.get("TIATOOLBOX_HOME"):
        roots.append(Path(os.environ["TIATOOLBOX_HOME"]) / "models")
    roots.append(Path.home() / ".tiatoolbox" / "models")
    for root in roots:
        cand = root / f"{model_name}.pth"
        if cand.is_file():
            return cand
    return None


def _create(arch: str, kwargs: dict):
    mod_name, cls_name = arch.split(".")
    import importlib

    mod = importlib.import_module(f"tiatoolbox_amd.models.architecture.{mod_name}")
    return getattr(mod, cls_name)(**kwargs)


def get_pretrained_model(pretrained_model: str | None = None, pretrained_weights: str | Path | None = None,
                         *, seed: int = 0):
    """Return ``(model, ioconfig)`` for a registered name (ref. :70-178)."""
    if not isinstance(pretrained_model, str):
        msg = "pretrained_model must be a string."
        raise TypeError(msg)
    if pretrained_model in ("resnet18", "resnet34", "resnet50", "resnet101"):
        # a bare torchvision backbone name -> feature extractor without ioconfig (ref. :133-134); seeded weights here
        from tiatoolbox_amd.models.architecture.vanilla import CNNBackbone

        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        try:
            model = CNNBackbone(pretrained_model)
        finally:
            torch.random.set_rng_state(gen_state)
        if pretrained_weights is not None:
            model.load_weights_from_file(pretrained_weights)
        return model, None
    if pretrained_model not in PRETRAINED_INFO:
        msg = f"Pretrained model `{pretrained_model}` does not exist."
        raise ValueError(msg)
    info = PRETRAINED_INFO[pretrained_model]
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = _create(*info["architecture"])
    finally:
        torch.random.set_rng_state(gen_state)
    if info.get("dataset"):
        model.preproc_func = predefined_preproc_func(info["dataset"])
    if pretrained_weights is None:
        pretrained_weights = local_pretrained_weights(pretrained_model)
    if pretrained_weights is not None:
        model.load_weights_from_file(pretrained_weights)
    else:
        logger.warning("No local weights for `%s` (the HuggingFace hub is unreachable): using the seeded "
                       "random initialisation. Pass `weights=<path to .pth>` for the pretrained model.",
                       pretrained_model)
    cfg_cls, cfg_kwargs = info["ioconfig"]
    return model, cfg_cls(**cfg_kwargs)
