"""Generate golden vectors by running the REAL reference (``/root/reference``) here.

Run in the build container only:  ``python tests/golden/make_golden.py``.
Third-party primitives that are absent (cv2 / skimage) are bound to the oracle's
restatements by ``_refshim``; everything else is the reference's own code.  Outputs are
small ``.npz`` fixtures committed next to this script; ``tests/test_oracle_golden.py``
checks the oracle against them and the ``-m gpu`` tests check the HIP path against both.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _refshim  # noqa: E402

_refshim.install()

from PIL import Image  # noqa: E402

from tiatoolbox_amd.utils import synth  # noqa: E402


def _ref_import(name: str):
    import importlib

    for attempt in range(3):  # the first import can trip over a lazy transformers import
        try:
            return importlib.import_module(name)
        except ModuleNotFoundError:
            if attempt == 2:
                raise
    return None


def _import_reference():
    import importlib

    for _ in range(2):  # first import can trip over a lazy transformers import
        try:
            return (importlib.import_module("tiatoolbox.tools.stainnorm"),
                    importlib.import_module("tiatoolbox.utils.misc"),
                    importlib.import_module("tiatoolbox.tools.stainaugment"))
        except ModuleNotFoundError:
            continue
    raise RuntimeError("cannot import the reference")


def stain_goldens() -> None:
    stainnorm, misc, stainaugment = _import_reference()
    target_full = np.array(Image.open(_refshim.REFERENCE / "tiatoolbox/data/target_image.png"))[..., :3]
    target = np.ascontiguousarray(target_full[:256, :256])
    np.save(HERE / "target_crop_256.npy", target)
    rng = np.random.default_rng(2)
    crops = []
    for _ in range(3):
        y, x = rng.integers(0, 1000 - 128, 2)
        crops.append(target_full[y:y + 128, x:x + 128])
    crops = np.ascontiguousarray(np.stack(crops))
    he = synth.g_he(3, 96, 96, seed=11)
    out = {"real_crops": crops, "he_seed": np.array(11)}

    for method in ("macenko", "ruifrok", "custom"):
        sm = np.array([[0.60, 0.72, 0.34], [0.10, 0.95, 0.29]]) if method == "custom" else None
        norm = stainnorm.get_normalizer(method, stain_matrix=sm)
        norm.fit(target.copy())
        out[f"{method}_stain_matrix_target"] = norm.stain_matrix_target
        out[f"{method}_maxC_target"] = norm.maxC_target
        out[f"{method}_stain_matrix_target_RGB"] = norm.stain_matrix_target_RGB
        out[f"{method}_target_conc_head"] = norm.target_concentrations[:64]
        out[f"{method}_real"] = np.stack([norm.transform(c.copy()) for c in crops])
        out[f"{method}_he"] = np.stack([norm.transform(c.copy()) for c in he])
        if method == "macenko":
            out["macenko_src_sm_real"] = np.stack([norm.extractor.get_stain_matrix(c.copy()) for c in crops])

    out["ce_real"] = np.stack([misc.contrast_enhancer(c.copy(), low_p=2, high_p=98) for c in crops])
    out["mask08_real"] = np.stack([misc.get_luminosity_tissue_mask(c.copy(), threshold=0.8) for c in crops])
    out["mask085_he"] = np.stack([misc.get_luminosity_tissue_mask(c.copy(), threshold=0.85) for c in he])

    # StainAugmentor with injected (alpha, beta) per stain channel (reference draws them unseeded)
    ab = np.array([[1.25, 0.9, 0.1, -0.05], [0.7, 1.3, -0.15, 0.12]])
    aug_out = []
    for k, (a0, a1, b0, b1) in enumerate(ab):
        seq = iter([(a0, b0), (a1, b1)])
        aug = stainaugment.StainAugmentor(method="macenko", sigma1=0.4, sigma2=0.2, augment_background=bool(k))

        def fake_params(self=aug, seq=seq):
            self.alpha, self.beta = next(seq)
            return {}

        aug.get_params = fake_params
        aug.fit(crops[k].copy(), threshold=0.85)
        aug_out.append(aug.augment())
    out["augment_ab"] = ab
    out["augment_real"] = np.stack(aug_out)
    np.savez_compressed(HERE / "stain_golden.npz", **out)
    print("wrote stain_golden.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


def mask_goldens() -> None:
    """OtsuTissueMasker / MorphologicalMasker of the real reference on real crops + synthetic patches."""
    tissuemask = _ref_import("tiatoolbox.tools.tissuemask")
    gold = np.load(HERE / "stain_golden.npz")
    crops = gold["real_crops"]
    he = synth.g_he(2, 160, 200, seed=31)
    out = {}
    for name, imgs in (("real", crops), ("he", he)):
        m = tissuemask.OtsuTissueMasker()
        out[f"otsu_{name}"] = m.fit_transform(imgs)
        out[f"otsu_thr_{name}"] = np.array(m.threshold)
        for tag, kw in (("k1", {"kernel_size": 1, "min_region_size": 6}), ("p125", {"power": 1.25}),
                        ("k5", {"kernel_size": 5}), ("mpp4", {"mpp": (4.0, 7.0)})):
            mm = tissuemask.MorphologicalMasker(**kw)
            out[f"morph_{tag}_{name}"] = mm.fit_transform(imgs)
            out[f"morph_{tag}_kernel"] = mm.kernel
            out[f"morph_{tag}_minreg"] = np.array(mm.min_region_size)
    np.savez_compressed(HERE / "mask_golden.npz", **out)
    print("wrote mask_golden.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


def hover_goldens() -> None:
    """HoVerNet._proc_np_hv / get_instance_info of the real reference on synthetic head outputs."""
    from oracle import hovernet as oh

    hov = _ref_import("tiatoolbox.models.architecture.hovernet")
    out = {}
    for tag, (h, w, seed, nb) in {"a": (164, 164, 1, 30), "b": (96, 120, 2, 12)}.items():
        npm, hv, tp = oh.synth_maps(2, h, w, seed=seed, n_blobs=nb)
        insts, boxes, cents, types, probs, ids, polys = [], [], [], [], [], [], []
        for i in range(2):
            inst = hov.HoVerNet._proc_np_hv(npm[i], hv[i])
            insts.append(inst)
            pred_type = np.around(tp[i]).astype("uint8")[..., 0]
            info = hov.HoVerNet.get_instance_info(inst, pred_type, verbose=False)
            ids.append(np.array(list(info.keys())))
            boxes.append(np.array([v["box"] for v in info.values()]))
            cents.append(np.array([v["centroid"] for v in info.values()]))
            types.append(np.array([v["type"] for v in info.values()]))
            probs.append(np.array([v["prob"] for v in info.values()]))
            polys.append([v["contours"] for v in info.values()])
        out[f"{tag}_shape"] = np.array([h, w, seed, nb])
        out[f"{tag}_inst"] = np.stack(insts)
        for i in range(2):
            out[f"{tag}_ids{i}"], out[f"{tag}_box{i}"], out[f"{tag}_cent{i}"] = ids[i], boxes[i], cents[i]
            out[f"{tag}_type{i}"], out[f"{tag}_prob{i}"] = types[i], probs[i]
            # polygons (through the shim's findContours = oracle.cvref.first_contour; cv2 itself is absent)
            out[f"{tag}_polylen{i}"] = np.array([len(c) for c in polys[i]])
            out[f"{tag}_poly{i}"] = np.concatenate(polys[i]).astype(np.int32)
    np.savez_compressed(HERE / "hover_golden.npz", **out)
    print("wrote hover_golden.npz:", {k: getattr(v, "shape", None) for k, v in out.items() if "inst" in k},
          [int(out[f"{t}_inst"].max()) for t in "ab"])


def hoverplus_goldens() -> None:
    """HoVerNet+ (hovernetplus.py): ``_proc_ls`` / ``_get_layer_info`` / ``_proc_np_hv(scale_factor=0.5)`` of the real
    reference on synthetic head outputs, and forward passes of the reference's own ``HoVerNet`` / ``HoVerNetPlus``
    modules carrying THIS repo's seeded weights (strict ``load_state_dict``), sub-sampled."""
    import torch

    from oracle import hovernet as oh
    from oracle import hovernetplus as ohp

    hp = _ref_import("tiatoolbox.models.architecture.hovernetplus")
    hov = _ref_import("tiatoolbox.models.architecture.hovernet")
    out = {}
    for tag, (h, w, seed) in {"a": (300, 340, 1), "b": (256, 256, 2)}.items():
        ls = ohp.synth_layer_map(h, w, seed=seed)
        layer = hp.HoVerNetPlus._proc_ls(ls)
        out[f"ls_{tag}_shape"] = np.array([h, w, seed])
        out[f"ls_{tag}_map"] = layer
        info = hp.HoVerNetPlus._get_layer_info(layer, (7, 3))
        out[f"ls_{tag}_type"] = np.array([int(v["type"]) for v in info.values()])
        out[f"ls_{tag}_box"] = np.array([v["box"] for v in info.values()])
        out[f"ls_{tag}_polylen"] = np.array([len(v["contours"]) for v in info.values()])
        out[f"ls_{tag}_poly"] = np.concatenate([v["contours"] for v in info.values()]).astype(np.int32)
    for tag, (h, w, seed, nb) in {"a": (164, 164, 41, 40), "b": (128, 150, 42, 25)}.items():
        npm, hv, _ = oh.synth_maps(2, h, w, seed=seed, n_blobs=nb)
        out[f"nuc_{tag}_shape"] = np.array([h, w, seed, nb])
        out[f"nuc_{tag}_inst"] = np.stack([hp.HoVerNetPlus._proc_np_hv(npm[i], hv[i], scale_factor=0.5) for i in range(2)])
    # forward parity: the reference modules with this repo's parameters
    from tiatoolbox_amd.models.architecture.hovernet import HoVerNet
    from tiatoolbox_amd.models.architecture.hovernetplus import HoVerNetPlus

    x = torch.from_numpy(synth.g_he(1, 256, 256, seed=77)).float().permute(0, 3, 1, 2)
    torch.manual_seed(5)
    mine = HoVerNet(num_types=6, mode="fast").eval()
    ref = hov.HoVerNet(num_types=6, mode="fast").eval()
    ref.load_state_dict(mine.state_dict(), strict=True)
    with torch.no_grad():
        o = ref(x)
    for k, v in o.items():
        out[f"fwd_hovernet_{k}"] = v[0, :, ::6, ::6].numpy()
    torch.manual_seed(6)
    mine = HoVerNetPlus(num_types=3, num_layers=5).eval()
    ref = hp.HoVerNetPlus(num_types=3, num_layers=5).eval()
    ref.load_state_dict(mine.state_dict(), strict=True)
    with torch.no_grad():
        o = ref(x)
    for k, v in o.items():
        out[f"fwd_hovernetplus_{k}"] = v[0, :, ::6, ::6].numpy()
    heads = hp.HoVerNetPlus.infer_batch(ref, x.permute(0, 2, 3, 1), device="cpu")
    out["fwd_hovernetplus_infer_tp"] = heads[2][0, ::4, ::4, 0]
    out["fwd_hovernetplus_infer_ls"] = heads[3][0, ::4, ::4, 0]
    np.savez_compressed(HERE / "hoverplus_golden.npz", **out)
    print("wrote hoverplus_golden.npz:", {k: v.shape for k, v in out.items() if "map" in k or "inst" in k or "fwd" in k})


def model_goldens() -> None:
    """Forward pass and ``infer_batch`` of the reference's own ``UNetModel`` (ResNet-50 encoder: decoder, skips, up-sampling are
    the real reference code; torchvision's ``ResNet`` / ``Bottleneck`` base classes are absent here and bound to the restatement
    in ``_refshim.bind_torchvision_resnet``) carrying THIS repo's seeded parameters (strict ``load_state_dict``), sub-sampled:
    pins the plain module the fused inference copy is tested against (``tests/test_model_forward_golden.py``)."""
    import torch

    _refshim.bind_torchvision_resnet()
    unet_ref = _ref_import("tiatoolbox.models.architecture.unet")
    from tiatoolbox_amd.models.architecture.unet import UNetModel

    out = {}
    x = torch.from_numpy(synth.g_he(1, 256, 256, seed=78)).float().permute(0, 3, 1, 2)
    torch.manual_seed(7)
    mine = UNetModel(3, 5, "resnet50", decoder_block=[3, 3]).eval()
    ref = unet_ref.UNetModel(3, 5, "resnet50", decoder_block=[3, 3]).eval()
    ref.load_state_dict(mine.state_dict(), strict=True)
    with torch.no_grad():
        o = ref(x)
    out["fwd_unet_shape"] = np.array(o.shape)
    out["fwd_unet"] = o[0, :, ::8, ::8].numpy()
    probs = unet_ref.UNetModel.infer_batch(ref, x.permute(0, 2, 3, 1), device="cpu")
    probs = probs[0] if isinstance(probs, (list, tuple)) else probs
    out["fwd_unet_infer_shape"] = np.array(np.asarray(probs).shape)
    out["fwd_unet_infer"] = np.asarray(probs)[0, ::8, ::8]
    torch.manual_seed(9)
    mine = UNetModel(3, 2, "unet", decoder_block=[3]).eval()  # the plain UNet encoder (fcn-tissue_mask layout)
    ref = unet_ref.UNetModel(3, 2, "unet", decoder_block=[3]).eval()
    ref.load_state_dict(mine.state_dict(), strict=True)
    with torch.no_grad():
        out["fwd_unet_plain"] = ref(x)[0, :, ::8, ::8].numpy()
    np.savez_compressed(HERE / "model_forward_golden.npz", **out)
    print("wrote model_forward_golden.npz:", {k: np.asarray(v).shape for k, v in out.items()})


def tile_goldens() -> None:
    """WSI tile-mode merge of instance predictions: the real reference's tile sets, margin rules, id stitching and
    offset handling (``multi_task_segmentor.py:1078-1287,1362-1554,2833-3297``) driven exactly like
    ``_process_tile_mode`` drives them, with the real ``HoVerNet.postproc`` per tile (shapely replaced by the
    axis-aligned stand-ins of ``oracle/geomref.py``)."""
    from types import SimpleNamespace

    from oracle import hovernet as oh

    mts = _ref_import("tiatoolbox.models.engine.multi_task_segmentor")
    hov = _ref_import("tiatoolbox.models.architecture.hovernet")
    ioc = _ref_import("tiatoolbox.models.engine.io_config")
    sem = _ref_import("tiatoolbox.models.engine.semantic_segmentor")
    out = {}
    res = {"units": "mpp", "resolution": 0.25}
    for tag, (rh, rw, seed, nb, tile, margin, pad) in {
            "a": (820, 750, 3, 420, 400, 32, (41, 82, 0, 0)),
            "b": (700, 1000, 4, 380, 500, 40, (0, 0, 0, 0))}.items():
        cfg = ioc.IOInstanceSegmentorConfig(input_resolutions=[res], output_resolutions=[res, res, res],
                                            patch_input_shape=[256, 256], patch_output_shape=[164, 164],
                                            stride_shape=[164, 164], margin=margin, tile_shape=[tile, tile])
        npm, hv, tp = oh.synth_maps(1, rh, rw, seed=seed, n_blobs=nb)
        heads = [npm[0], hv[0], tp[0]]
        wsi_shape = (rw + pad[0] + pad[2], rh + pad[1] + pad[3])
        fake = SimpleNamespace(_ioconfig=cfg, mask_padding=pad,
                               dataloader=SimpleNamespace(dataset=SimpleNamespace(mask_reader=None)))
        sets = mts.MultiTaskSegmentor._get_tile_info(fake, image_shape=(rw, rh), wsi_proc_shape=wsi_shape)
        meta = mts._build_tile_tasks(tile_info_sets=sets, verbose=False)
        base = cfg.to_baseline()
        model_self = SimpleNamespace(tasks=["nuclei_segmentation"])
        wsi_info, max_inst = None, None
        for bounds, flag, mode in meta:
            crop = [p[bounds[1]:bounds[3], bounds[0]:bounds[2], :] for p in heads]
            post = hov.HoVerNet.postproc(model_self, crop)
            if wsi_info is None:
                wsi_info = ({"task_type": post[0]["task_type"], "info_dict": {},
                             "predictions": np.zeros(wsi_shape[::-1], dtype=post[0]["predictions"].dtype)},)
            wsi_info, max_inst = mts._update_tile_based_predictions_array(
                post_process_output=post, wsi_info_dict=wsi_info, bounds=bounds, offset=(pad[0], pad[1]),
                max_inst_value=max_inst)
            tl, br = bounds[:2], bounds[2:]
            for k, inst_dict in enumerate(mts._get_inst_info_dicts(post_process_output=post)):
                fresh, stale = mts._compute_info_dict_for_merge(
                    inst_dict=inst_dict, tile_mode=mode, ref_inst_info_dict=wsi_info[k]["info_dict"], ioconfig=base,
                    tile_shape=br - tl, tile_tl=tl, tile_flag=flag)
                wsi_info[k]["info_dict"].update(fresh)
                for key in stale:
                    wsi_info[k]["info_dict"].pop(key, None)
        recs = list(wsi_info[0]["info_dict"].values())
        cols = {key: mts.apply_coordinate_offset(
            data_array=np.array([r[key] for r in recs] + [None], dtype=object)[:-1], offset=np.array(pad[:2]), key=key,
            verbose=False) for key in ("box", "centroid", "contours")}
        out[f"{tag}_cfg"] = np.array([rh, rw, seed, nb, tile, margin, *pad])
        for si, (b, f) in enumerate(sets):
            out[f"{tag}_set{si}_bounds"], out[f"{tag}_set{si}_flags"] = np.asarray(b), np.asarray(f)
        out[f"{tag}_box"] = np.array([c for c in cols["box"]]).reshape(-1, 4)
        out[f"{tag}_centroid"] = np.array([c for c in cols["centroid"]]).reshape(-1, 2)
        out[f"{tag}_polylen"] = np.array([len(c) for c in cols["contours"]])
        out[f"{tag}_poly"] = np.concatenate(list(cols["contours"])).astype(np.int32)
        out[f"{tag}_type"] = np.array([r["type"] for r in recs])
        out[f"{tag}_prob"] = np.array([r["prob"] for r in recs])
        out[f"{tag}_pred"] = wsi_info[0]["predictions"]
        print(tag, "tiles per set", [len(b) for b, _ in sets], "instances", len(recs), "max id", int(wsi_info[0]["predictions"].max()))
    # tile sets under a tissue mask (tiles whose box holds no mask pixel are dropped before the seam sets are built)
    wsr = _ref_import("tiatoolbox.wsicore.wsireader")
    rng = np.random.default_rng(9)
    for k, (rw, rh, tile, margin, pad) in enumerate([(2600, 2100, 700, 64, (120, 60, 0, 0)), (1900, 3000, 1024, 128, (0, 0, 0, 0))]):
        cfg = ioc.IOInstanceSegmentorConfig(input_resolutions=[res], output_resolutions=[res, res, res],
                                            patch_input_shape=[256, 256], patch_output_shape=[164, 164],
                                            stride_shape=[164, 164], margin=margin, tile_shape=[tile, tile])
        wsi_shape = (rw + pad[0], rh + pad[1])
        mask = (rng.random((wsi_shape[1] // 32, wsi_shape[0] // 32)) < 0.012).astype(np.uint8)
        reader = object.__new__(wsr.VirtualWSIReader)
        reader.img = mask
        fake = SimpleNamespace(_ioconfig=cfg, mask_padding=pad,
                               dataloader=SimpleNamespace(dataset=SimpleNamespace(mask_reader=reader)))
        sets = mts.MultiTaskSegmentor._get_tile_info(fake, image_shape=(rw, rh), wsi_proc_shape=wsi_shape)
        out[f"masked{k}_cfg"] = np.array([rw, rh, tile, margin, *pad])
        out[f"masked{k}_mask"] = mask
        for si, (b, f) in enumerate(sets):
            out[f"masked{k}_set{si}_bounds"], out[f"masked{k}_set{si}_flags"] = np.asarray(b).reshape(-1, 4), np.asarray(f).reshape(-1, 4)
        print("masked", k, "tiles per set", [len(b) for b, _ in sets])
    # region bookkeeping of infer_wsi
    rng = np.random.default_rng(0)
    for k in range(4):
        h, w = int(rng.integers(700, 1500)), int(rng.integers(700, 1500))
        xs, ys = np.arange(0, w, 164), np.arange(0, h, 164)
        locs = np.array([[x, y, x + 164, y + 164] for y in ys for x in xs])
        keep = rng.random(len(locs)) < 0.3
        kept = locs[keep]
        mb = (kept[:, 0].min(), kept[:, 1].min(), kept[:, 2].max(), kept[:, 3].max())
        inside, padding, shape = sem.get_full_output_locs_inside_mask(full_output_locs=locs.copy(), mask_bounds=mb,
                                                                      output_shape=(h, w))
        out[f"region{k}_in"] = np.array([h, w, *mb])
        out[f"region{k}_locs"], out[f"region{k}_inside"] = locs, inside
        out[f"region{k}_padding"], out[f"region{k}_shape"] = np.array(padding), np.array(shape)
    np.savez_compressed(HERE / "tile_golden.npz", **out)
    print("wrote tile_golden.npz")


def grid_goldens() -> None:
    """PatchExtractor.get_coordinates / filter_coordinates and merge_batch_to_canvas of the real reference."""
    pe = _ref_import("tiatoolbox.tools.patchextraction")
    wsr = _ref_import("tiatoolbox.wsicore.wsireader")
    out = {}
    cases = {"a": ((2000, 1500), (1024, 1024), (512, 512), (450, 450)), "b": ((333, 777), (64, 48), (32, 24), (20, 17)),
             "c": ((500, 500), (100, 100), (100, 100), (100, 100))}
    rng = np.random.default_rng(5)
    for tag, (img, pin, pout, stride) in cases.items():
        i_b, o_b = pe.PatchExtractor.get_coordinates(patch_output_shape=pout, image_shape=img, patch_input_shape=pin,
                                                     stride_shape=stride)
        out[f"{tag}_args"] = np.array([img, pin, pout, stride])
        out[f"{tag}_in"], out[f"{tag}_out"] = i_b, o_b
        mask = (rng.random((max(img[1] // 16, 4), max(img[0] // 16, 4))) < 0.35).astype(np.uint8)
        reader = object.__new__(wsr.VirtualWSIReader)
        reader.img = mask
        out[f"{tag}_mask"] = mask
        for ratio in (0.0, 0.4):
            out[f"{tag}_keep{int(ratio * 10)}"] = pe.PatchExtractor.filter_coordinates(reader, o_b, img, min_mask_ratio=ratio)
    ss = _ref_import("tiatoolbox.models.engine.semantic_segmentor")
    blocks = rng.random((5, 16, 16, 3)).astype(np.float32)
    blocks[2] = 0
    locs = np.array([[0, 0, 16, 16], [12, 0, 28, 16], [24, 0, 40, 16], [36, 0, 52, 16], [48, 0, 64, 16]])
    locs[:, 2] = np.minimum(locs[:, 2], 60)
    canvas, count = ss.merge_batch_to_canvas(blocks, locs, (16, 60, 3))
    out["merge_blocks"], out["merge_locs"], out["merge_canvas"], out["merge_count"] = blocks, locs, canvas, count
    np.savez_compressed(HERE / "grid_golden.npz", **out)
    print("wrote grid_golden.npz", len(out))


def reinhard_goldens() -> None:
    """ReinhardNormalizer of the real reference (cv2 colour conversions bound to the oracle's restatements)."""
    stainnorm, _, _ = _import_reference()
    gold = np.load(HERE / "stain_golden.npz")
    target = np.load(HERE / "target_crop_256.npy")
    crops = gold["real_crops"]
    he = synth.g_he(3, 96, 96, seed=int(gold["he_seed"]))
    norm = stainnorm.get_normalizer("reinhard")
    norm.fit(target.copy())
    out = {"target_means": np.array(norm.target_means), "target_stds": np.array(norm.target_stds),
           "real": np.stack([norm.transform(c.copy()) for c in crops]), "he": np.stack([norm.transform(c.copy()) for c in he])}
    ms = [norm.get_mean_std(c.copy()) for c in crops]
    out["real_means"] = np.array([m[0] for m in ms])
    out["real_stds"] = np.array([m[1] for m in ms])
    np.savez_compressed(HERE / "reinhard_golden.npz", **out)
    print("wrote reinhard_golden.npz", out["target_means"], out["target_stds"])


if __name__ == "__main__":
    which = sys.argv[1:] or ["stain", "mask", "hover", "hoverplus", "grid", "reinhard", "tile", "models"]
    if "hoverplus" in which:
        hoverplus_goldens()
    if "models" in which:
        model_goldens()
    if "reinhard" in which:
        reinhard_goldens()
    if "grid" in which:
        grid_goldens()
    if "hover" in which:
        hover_goldens()
    if "tile" in which:
        tile_goldens()
    if "stain" in which:
        stain_goldens()
    if "mask" in which:
        mask_goldens()
