"""Multi-head segmentation engines in patch mode (API of reference
``tiatoolbox/models/engine/multi_task_segmentor.py`` and ``nucleus_instance_segmentor.py``).

``infer_patches`` keeps every head's output resident on the GPU; ``post_process_patches`` runs the
model's *batched* device post-processing (HoVer-Net: Sobel/energy/CCL/watershed kernels over all
patches of a chunk at once) instead of a Python loop over patches.

WSI mode (reference :477-731, 836-1554, 2833-3297): every head is stitched into one device-resident map of
the tissue region; maps no larger than ``ioconfig.tile_shape`` are post-processed in one go, larger ones tile
by tile (tiles of equal shape batched through the device pipeline) and merged across the seams with the
reference's four tile sets / margin rules.  Results are plain NumPy containers (no dask / zarr).
"""

from __future__ import annotations

import uuid
import warnings

import numpy as np
import torch

from tiatoolbox_amd.utils import tracing

from tiatoolbox_amd.models.engine.engine_abc import EngineABC
from tiatoolbox_amd.tools.patchextraction import PatchExtractor
from tiatoolbox_amd.wsicore import ArrayWSIReader


# ===================================================================================== WSI mode
# Tile-mode merging of instance predictions (reference :1078-1287, 1362-1554, 2833-3297).  The reference
# expresses every test with shapely boxes and an STRtree; all of them are axis-aligned rectangle
# predicates, evaluated here directly on NumPy box arrays:
#   STRtree.query(box)     -> rectangles whose closed extents meet (touching counts)
#   box.contains(instance) -> the instance rectangle lies inside a non-degenerate box
def _boxes_meeting(boxes: np.ndarray, q) -> np.ndarray:
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    return np.flatnonzero((b[:, 0] <= q[2]) & (b[:, 2] >= q[0]) & (b[:, 1] <= q[3]) & (b[:, 3] >= q[1]))


def _box_holds(q, box) -> bool:
    return bool(q[2] > q[0] and q[3] > q[1] and q[0] <= box[0] and q[1] <= box[1] and q[2] >= box[2] and q[3] >= box[3])


def get_full_output_locs_inside_mask(full_output_locs: np.ndarray, mask_bounds, output_shape):
    """Output locations that meet the tissue bounding box, re-based to its top-left patch; the padding that
    puts the region back into the slide; the region's (height, width) (ref. ``semantic_segmentor.py:1757-1802``)."""
    locs = np.asarray(full_output_locs)
    mx0, my0, mx1, my1 = mask_bounds
    meets = ~((locs[:, 2] < mx0) | (locs[:, 0] > mx1) | (locs[:, 3] < my0) | (locs[:, 1] > my1))
    inside = locs[meets].copy()
    min_x, min_y = int(inside[:, 0].min()), int(inside[:, 1].min())
    max_x, max_y = int(inside[:, 2].max()), int(inside[:, 3].max())
    pad_bottom, pad_right = max(int(output_shape[0]) - max_y, 0), max(int(output_shape[1]) - max_x, 0)
    inside -= np.array([min_x, min_y, min_x, min_y])
    region = (int(output_shape[0]) - min_y - pad_bottom, int(output_shape[1]) - min_x - pad_right)
    return inside, (min_x, min_y, pad_right, pad_bottom), region


def apply_coordinate_offset(data_array: np.ndarray, offset, key, keys_to_shift=("centroid", "box", "contours")):
    """Shift boxes (4-vectors) by (dx, dy, dx, dy) and points / polygons by (dx, dy) (ref. :3760-3829)."""
    dx, dy = offset
    if key not in keys_to_shift or (dx == 0 and dy == 0):
        return data_array
    out = np.empty(len(data_array), dtype=object)
    for i, item in enumerate(data_array):
        is_box = item.ndim == 1 and item.size == 4  # noqa: PLR2004
        out[i] = (item + (np.array([dx, dy, dx, dy]) if is_box else np.array([dx, dy]))).astype(item.dtype)
    return out


def _tile_frames(width: int, height: int, margin: int):
    """Per side (top, bottom, left, right): the margin band and the one-pixel boundary band of a tile."""
    bands = [(0, 0, width, margin), (0, height - margin, width, height), (0, 0, margin, height),
             (width - margin, 0, width, height)]
    edges = [(0, 0, width, 1), (0, height - 1, width, height), (0, 0, 1, height), (width - 1, 0, width, height)]
    return bands, edges


def _get_margin_lines(margin: int, height: int, width: int, tile_tl) -> list:
    """The four inner margin lines of a tile in slide coordinates, as degenerate rectangles (ref. :3023-3038)."""
    tx, ty = int(tile_tl[0]), int(tile_tl[1])
    lines = [(margin, margin, width - margin, margin), (margin, height - margin, width - margin, height - margin),
             (margin, margin, margin, height - margin), (width - margin, margin, width - margin, height - margin)]
    return [(a + tx, b + ty, c + tx, d + ty) for a, b, c, d in lines]


def _get_sel_indices_margin_lines(ioconfig, tile_shape, tile_flag, tile_mode: int, tile_tl, inst_dict: dict):
    """Indices of the tile's instances to discard, and the tile's margin lines (ref. :2944-3020).

    Grid tiles (mode 0) and cross tiles (mode 3) drop instances that lie wholly inside a flagged margin band
    (every band for mode 3); strip tiles (modes 1, 2) drop everything that meets the flagged sides' margin
    band, or the boundary band of the unflagged sides.
    """
    if tile_mode not in (0, 1, 2, 3):
        msg = f"Unknown tile mode {tile_mode}."
        raise ValueError(msg)
    margin = 0 if ioconfig.margin is None else int(ioconfig.margin)
    width, height = (int(v) for v in tile_shape)
    boxes = np.array([v["box"] for v in inst_dict.values()])
    bands, edges = _tile_frames(width, height, margin)
    lines = _get_margin_lines(margin, height, width, tile_tl)
    picked: list[int] = []
    if tile_mode in (0, 3):
        for side, band in enumerate(bands):
            if tile_flag[side] or tile_mode == 3:  # noqa: PLR2004
                picked += [int(i) for i in _boxes_meeting(boxes, band) if _box_holds(band, boxes[i])]
        return picked, lines
    for side, flagged in enumerate(tile_flag):
        picked += [int(i) for i in _boxes_meeting(boxes, bands[side] if flagged else edges[side])]
    return picked, lines


def retrieve_sel_uids(sel_indices: list, inst_dict: dict) -> list:
    keys = list(inst_dict.keys()) if len(sel_indices) else []
    return [keys[i] for i in sel_indices]


def _move_tile_space_to_wsi_space(inst_dict: dict, tile_tl, remove_insts_in_tile: list) -> dict:
    """Surviving instances, shifted to slide coordinates, under fresh unique keys (ref. :3041-3058)."""
    dropped = set(remove_insts_in_tile)
    shift = np.asarray(tile_tl)
    moved = {}
    for key, info in inst_dict.items():
        if key in dropped:
            continue
        info["box"] = info["box"] + np.concatenate([shift, shift])
        if "centroid" in info:
            info["centroid"] = info["centroid"] + shift
        info["contours"] = info["contours"] + shift.astype(info["contours"].dtype)
        moved[uuid.uuid4().hex] = info
    return moved


def _compute_info_dict_for_merge(inst_dict: dict, tile_mode: int, ref_inst_info_dict: dict, ioconfig, tile_shape,
                                 tile_tl, tile_flag):
    """``(new instances in slide space, keys of accumulated instances to delete)`` for one tile (ref. :2833-2933,
    3269-3296): cross tiles (mode 3) replace whatever already sits across their margin lines."""
    if len(inst_dict) == 0:
        return {}, []
    picked, lines = _get_sel_indices_margin_lines(ioconfig, tile_shape, tile_flag, tile_mode, tile_tl, inst_dict)
    moved = _move_tile_space_to_wsi_space(inst_dict, tile_tl, retrieve_sel_uids(picked, inst_dict))
    if tile_mode != 3 or not ref_inst_info_dict:  # noqa: PLR2004
        return moved, []
    ref_boxes = np.array([v["box"] for v in ref_inst_info_dict.values()])
    crossing = [int(i) for line in lines for i in _boxes_meeting(ref_boxes, line)]
    return moved, retrieve_sel_uids(crossing, ref_inst_info_dict)


def _get_inst_info_dicts(post_process_output) -> list:
    """Column-wise ``info_dict`` of every task -> ``{1-based id: {key: value}}`` (ref. :3061-3083)."""
    dicts = []
    for out in post_process_output:
        cols = out["info_dict"]
        first = next(iter(cols))
        dicts.append({i + 1: {k: v[i] for k, v in cols.items()} for i in range(len(cols[first]))})
    return dicts


def _update_tile_based_predictions_array(post_process_output, wsi_info_dict, bounds, offset, max_inst_value=None):
    """Paste a tile's prediction map into the slide map; instance ids are shifted by the running maximum and
    instances overlapping something already present are left out (ref. :3160-3212)."""
    b = np.array(bounds)
    b[:2] += np.asarray(offset)
    b[2:] += np.asarray(offset)
    x0, y0, x1, y1 = (int(v) for v in b)
    for idx, out in enumerate(post_process_output):
        full = wsi_info_dict[idx]["predictions"]
        if full is None:
            continue
        x1, y1 = min(x1, full.shape[1]), min(y1, full.shape[0])
        new = out["predictions"][0:y1 - y0, 0:x1 - x0]
        if out["seg_type"] == "instance":
            old = full[y0:y1, x0:x1]
            clash = (new > 0) & (old > 0)
            max_inst_value = 0 if max_inst_value is None else max_inst_value
            keep = new > 0
            if clash.any():
                bad = np.unique(new[clash])
                keep &= ~np.isin(new, bad[bad > 0])
            merged = old.copy()
            merged[keep] = new[keep] + max_inst_value
            if keep.any():
                max_inst_value = merged.max() + max_inst_value
            new = merged
        full[y0:y1, x0:x1] = new
    return wsi_info_dict, max_inst_value


def _build_tile_tasks(tile_info_sets: list) -> list:
    return [(bounds, flags[i], mode) for mode, (all_bounds, flags) in enumerate(tile_info_sets)
            for i, bounds in enumerate(all_bounds)]


def _sides_on_region_border(boxes: np.ndarray, flags: np.ndarray, w: int, h: int) -> np.ndarray:
    """Clear the flag of every tile side that lies on the border of the processed region."""
    borders = [(0, 0, w, 0), (0, h, w, h), (0, 0, 0, h), (w, 0, w, h)]
    for side, line in enumerate(borders):
        flags[_boxes_meeting(boxes, line), side] = 0
    return flags


class MultiTaskSegmentor(EngineABC):
    """Multi-task (here: nuclei instance) segmentation in patch and WSI mode (ref. :229-3829)."""

    def __init__(self, model, batch_size: int = 8, num_workers: int = 0, weights=None, *, device: str = "cpu",
                 verbose: bool = True) -> None:
        super().__init__(model=model, batch_size=batch_size, num_workers=num_workers, weights=weights, device=device,
                         verbose=verbose)
        self.return_probabilities = False
        self.mask_bounds = None
        self.mask_padding = (0, 0, 0, 0)
        self.fold_batchnorm = False  # HoVer-Net's pre-activation BN->ReLU->conv order cannot be folded forwards
        # multi-process runs: every rank post-processes its own patch shard and the instance tables are gathered
        #  instead of gathering the raw head maps and post-processing everything everywhere
        self.gather_raw_predictions = False
        self.tasks = set(getattr(self.model, "tasks", []))

    def _update_run_params(self, images, **kwargs):
        if kwargs.get("return_labels"):
            msg = "`return_labels` is not supported for MultiTaskSegmentor."
            raise ValueError(msg)  # ref. :2149-2156
        self.return_probabilities = bool(kwargs.get("return_probabilities", False))  # per call (ref. :982, 1710, 1824)
        return super()._update_run_params(images, **kwargs)

    def post_process_patches(self, raw_predictions: dict, **_) -> dict:
        """Per-patch ``postproc`` for every patch, batched on the device (ref. :733-834, :1556-1685)."""
        heads = raw_predictions["probabilities"]
        model = self.model.module if hasattr(self.model, "module") else self.model
        if len(getattr(model, "tasks", ())) > 1:
            return self._post_process_patches_multi_task(raw_predictions, model)
        results: list[dict] = []
        n = heads[0].shape[0]
        on_gpu = heads[0].is_cuda
        chunk = 2048
        for s in range(0, n, chunk):
            part = [h[s:s + chunk] for h in heads]
            if on_gpu and hasattr(model, "postproc_batch"):
                with tracing.range("hover_postproc_batch"):
                    results += model.postproc_batch(part[0], part[1], part[2] if len(part) > 2 else None)  # noqa: PLR2004
            else:
                postproc = self._get_model_attr("postproc_func")
                for i in range(part[0].shape[0]):
                    results.append(postproc([p[i].cpu().numpy() for p in part], offset=(0, 0))[0])
        out: dict = {}
        task = results[0]["task_type"] if results else "nuclei_segmentation"
        self.tasks = {task}
        preds = (np.stack([r["predictions"] for r in results]) if results else np.empty((0,), dtype=np.int32))
        tables = [r["info_dict"] for r in results]
        if "shard" in raw_predictions:  # patch-sharded run: gather label maps and instance tables in input order
            from tiatoolbox_amd import distributed as tdist

            dev = heads[0].device if isinstance(heads[0], torch.Tensor) else torch.device("cpu")
            _, _, n_total = raw_predictions["shard"]
            side = tuple(heads[0].shape[1:3])
            local = torch.from_numpy(np.ascontiguousarray(preds.reshape(-1, *side).astype(np.int32))).to(dev)
            preds = tdist.all_gather_rows(local, n_total).cpu().numpy()
            tables = tdist.gather_instance_tables(tables, dev)
            if self.return_probabilities:
                heads = [tdist.all_gather_rows(h if isinstance(h, torch.Tensor) else torch.from_numpy(np.asarray(h)), n_total)
                         for h in heads]
        out["predictions"] = preds
        for key in ("box", "centroid", "contours", "prob", "type"):
            out[key] = [t[key] for t in tables]
        if self.return_probabilities:
            out["probabilities"] = [h.cpu().numpy() if isinstance(h, torch.Tensor) else h for h in heads]
        return out

    # ------------------------------------------------------------------------------ WSI mode
    def _get_tile_info(self, image_shape, wsi_proc_shape, mask_reader=None) -> list:
        """Four tile sets with their removal flags (ref. :1362-1554): the regular grid, vertical seam strips,
        horizontal seam strips and seam crossings.  ``image_shape`` = (width, height) of the processed region."""
        cfg = self._ioconfig
        margin = 0 if cfg.margin is None else int(cfg.margin)
        out_shape = np.asarray(cfg.patch_output_shape)
        tile_shape = (np.floor(np.asarray(cfg.tile_shape) / out_shape) * out_shape).astype(np.int32)
        image_shape = np.asarray(image_shape)
        boxes = PatchExtractor.get_coordinates(image_shape=image_shape, patch_input_shape=tile_shape,
                                               patch_output_shape=tile_shape, stride_shape=tile_shape)[1]
        shift = np.array([*self.mask_padding[:2], *self.mask_padding[:2]])
        if mask_reader is not None:
            boxes = boxes[PatchExtractor.filter_coordinates(mask_reader, boxes + shift, wsi_shape=wsi_proc_shape,
                                                            min_mask_ratio=0)]
        if np.all(image_shape <= tile_shape):
            return [[boxes, np.zeros((len(boxes), 4), dtype=np.int32)]]
        w, h = (int(v) for v in image_shape)
        flag = _sides_on_region_border(boxes, np.ones((len(boxes), 4), dtype=np.int32), w, h)
        sets = [[boxes, flag]]
        bottom_right, top_right, bottom_left = boxes[:, 2:], boxes[:, [2, 1]], boxes[:, [0, 3]]
        # vertical strips over the right seams
        pick = np.flatnonzero(flag[:, 3])
        strips = np.concatenate([top_right[pick] - [margin, 0], bottom_right[pick] + [margin, 0]], axis=-1)
        sflag = np.zeros((len(strips), 4), dtype=np.int32)
        sflag[:, [0, 1]] = 1
        sets.append([strips, _sides_on_region_border(strips, sflag, w, h)])
        # horizontal strips over the bottom seams
        pick = np.flatnonzero(flag[:, 1])
        strips = np.concatenate([bottom_left[pick] - [0, margin], bottom_right[pick] + [0, margin]], axis=-1)
        sflag = np.zeros((len(strips), 4), dtype=np.int32)
        sflag[:, [2, 3]] = 1
        sets.append([strips, _sides_on_region_border(strips, sflag, w, h)])
        # squares over the seam crossings
        pick = np.flatnonzero(flag[:, 1] * flag[:, 3])
        squares = np.concatenate([bottom_right[pick] - 2 * margin, bottom_right[pick] + 2 * margin], axis=-1)
        sets.append([squares, np.ones((len(squares), 4), dtype=np.int32)])
        return sets

    def infer_wsi(self, reader: ArrayWSIReader, mask_reader: ArrayWSIReader | None = None) -> dict:
        """Patch inference over the tissue region of one slide, every head stitched into one device map
        (ref. :477-731).  Sets ``mask_bounds`` / ``mask_padding`` like the reference."""
        from tiatoolbox_amd import _lib
        from tiatoolbox_amd.models.engine.engine_abc import _DTYPES
        from tiatoolbox_amd.models.engine.semantic_segmentor import _finalize, _row_merge

        dev = torch.device(self.device)
        if dev.type != "cuda":
            msg = "WSI-mode stitching runs on the GPU (device='cuda'); there is no CPU fallback."
            raise _lib.HipLibraryError(msg)
        cfg = self._ioconfig
        w, h = reader.slide_dimensions
        in_b, out_b = PatchExtractor.get_coordinates(
            patch_output_shape=tuple(cfg.patch_output_shape[::-1]), image_shape=(w, h),
            patch_input_shape=tuple(cfg.patch_input_shape[::-1]), stride_shape=tuple(cfg.stride_shape[::-1]))
        keep = np.ones(len(in_b), dtype=bool)
        if mask_reader is not None:
            keep = PatchExtractor.filter_coordinates(mask_reader, out_b, (w, h), min_mask_ratio=0)
        if not keep.any():
            self.mask_bounds, self.mask_padding = None, (0, 0, 0, 0)
            return {"probabilities": None, "coordinates": out_b[keep]}
        kept = out_b[keep]
        self.mask_bounds = (kept[:, 0].min(), kept[:, 1].min(), kept[:, 2].max(), kept[:, 3].max())
        inside, self.mask_padding, (rh, rw) = get_full_output_locs_inside_mask(out_b, self.mask_bounds, (h, w))
        min_x, min_y = self.mask_padding[:2]
        dtype = _DTYPES[str(self.compute_dtype).replace("torch.", "")]
        model = self._inference_model(dtype)
        infer_batch = self._get_model_attr("infer_batch")
        oh = int(cfg.patch_output_shape[0])
        heads: list[torch.Tensor] | None = None
        prev: list | None = None   # per head (row, cnt, ys)
        row_ys = np.unique(inside[:, 1])
        # patch rows are sharded over ranks: every rank stitches the band of head-map rows it owns (plus one leading
        # patch row for the overlap) and the bands are all-gathered once; one rank = the whole region
        from tiatoolbox_amd import distributed as tdist
        from tiatoolbox_amd.models.engine.semantic_segmentor import band_plan, exchange_bands

        rank, world = tdist.world() if self.distributed else (0, 1)
        plan = band_plan(row_ys, oh, rh, rank, world)
        r_lo = plan["own"][0]
        y_lo, y_hi = plan["y_lo"], plan["y_hi"]
        band_h = max(y_hi - y_lo, 0)
        band = None  # CanvasBand: the heads' maps resident on the device, or streamed to page-locked host memory (slides > HBM)

        def open_band(channels: list[int]):
            from tiatoolbox_amd.models.engine.semantic_segmentor import CanvasBand, _band_device_rows, exchange_footprint

            # "pred" = the arg-max plane the finalize kernel always writes (not used by this engine); in streamed mode it exists per
            # chunk only
            maps = {f"h{j}": ((c,), torch.float32) for j, c in enumerate(channels)}
            k = _band_device_rows(self, CanvasBand.bytes_needed(band_h, rw, maps), dev, world=world,
                                  exchange_bytes=exchange_footprint(plan, rh, rw, maps))
            cb = CanvasBand(band_h, rw, y_lo, oh, dev, maps, device_rows=k)
            rows_ = oh if cb.streamed else band_h
            cb.scratch_pred = [torch.zeros((rows_, rw), dtype=torch.uint8, device=dev) for _ in range(cb.k if cb.streamed else 1)]
            return cb

        from tiatoolbox_amd.models.engine.engine_abc import iter_row_outputs

        plan_rows = list(plan["rows"])
        row_sels = [np.flatnonzero(keep & (out_b[:, 1] - min_y == int(row_ys[ri]))) for ri in plan_rows]

        # bounds and row offsets to the device once, in the order `iter_row_outputs` consumes them (see SemanticSegmentor.infer_wsi)
        flat = np.concatenate([s for s in row_sels if len(s)]) if any(len(s) for s in row_sels) else np.zeros(0, np.int64)
        bs = int(self.batch_size)
        size = (int(in_b[0, 2] - in_b[0, 0]), int(in_b[0, 3] - in_b[0, 1])) if len(in_b) else (0, 0)
        uniform = len(in_b) > 0 and bool(np.all(in_b[:, 2] - in_b[:, 0] == size[0]) and np.all(in_b[:, 3] - in_b[:, 1] == size[1]))
        bounds_dev = xs_dev = None
        if uniform and len(flat):
            padded = np.concatenate([flat, np.repeat(flat[-1:], bs)])
            bounds_dev = torch.from_numpy(np.ascontiguousarray(in_b[padded], dtype=np.int32)).to(dev)
            xs_dev = torch.from_numpy(np.ascontiguousarray(out_b[flat, 0] - min_x, dtype=np.int32)).to(dev)
        row_starts = np.concatenate([[0], np.cumsum([len(s) for s in row_sels])])
        state = {"pos": 0}

        def infer(idx):
            if bounds_dev is None:
                return tuple(infer_batch(model, reader.read_bounds_batch(in_b[idx]), device=self.device))
            p = state["pos"]
            state["pos"] = p + min(bs, len(flat) - p)
            return tuple(infer_batch(model, reader.read_bounds_batch(bounds_dev[p:p + bs], size=size), device=self.device))

        with self._miopen_scope():
            for k, outs in iter_row_outputs(infer, row_sels, self.batch_size):
                ri, sel = plan_rows[k], row_sels[k]
                ys = int(row_ys[ri])
                rows = None
                if outs is not None:
                    blocks = [o.float().contiguous() for o in outs]
                    if heads is None:
                        band = open_band([b.shape[-1] for b in blocks])
                        heads = [band.full[f"h{j}"] for j in range(len(blocks))]
                    xs = xs_dev[int(row_starts[k]):int(row_starts[k + 1])] if xs_dev is not None else out_b[sel, 0] - min_x
                    rows = [(*_row_merge(b, xs, rw), ys) for b in blocks]
                if heads is None:
                    continue  # nothing inferred yet: the maps start as zeros
                if rows is None:
                    rows = [(torch.zeros((oh, rw, hd.shape[-1]), dtype=torch.float32, device=dev),
                             torch.zeros((oh, rw), dtype=torch.uint8, device=dev), ys) for hd in heads]
                if ri >= r_lo:
                    # the band [ys, next row): this row plus whatever the previous row still covers
                    y1 = min(int(row_ys[ri + 1]) if ri + 1 < len(row_ys) else ys + oh, rh)
                    dst, y_base = band.target(ys)
                    dummy = band.scratch_pred[band._slot if band.streamed else 0]  # noqa: SLF001
                    for j in range(len(heads)):
                        cur = rows[j]
                        if prev is None:
                            _finalize(cur[0], cur[1], cur[2], None, None, 0, ys, y1, dst[f"h{j}"], dummy, y_base=y_base)
                        else:
                            _finalize(prev[j][0], prev[j][1], prev[j][2], cur[0], cur[1], cur[2], ys, y1, dst[f"h{j}"], dummy,
                                      y_base=y_base)
                    band.done(ys, y1)
                prev = rows
        if band is not None:
            maps = band.result()  # streamed: waits for the last copies; the maps are host tensors then
            heads = [maps[f"h{j}"] for j in range(len(heads))]
        self.last_band_streamed = bool(band is not None and band.streamed)
        if world > 1:
            if heads is None:  # this rank's rows hold no tissue: it still takes part in the exchange with zero bands
                probe = infer_batch(model, reader.read_bounds_batch(in_b[keep][:1]), device=self.device)
                band = open_band([p.shape[-1] for p in probe])
                heads = [band.result()[f"h{j}"] for j in range(len(probe))]
            if band.streamed:
                from tiatoolbox_amd.models.engine.semantic_segmentor import exchange_bands_streamed

                heads = [exchange_bands_streamed(hd, plan, rh, dev) for hd in heads]
            else:
                heads = [exchange_bands(hd, plan, rh) for hd in heads]
        return {"probabilities": heads, "coordinates": kept}

    def _postproc_maps(self, maps: list[torch.Tensor], offset=(0, 0)) -> tuple[dict, ...]:
        """The model's ``postproc`` on device-resident head maps; ``predictions`` come back as NumPy."""
        model = self.model.module if hasattr(self.model, "module") else self.model
        out = model.postproc(list(maps), offset=offset)
        for task in out:
            if isinstance(task.get("predictions"), torch.Tensor):
                task["predictions"] = task["predictions"].cpu().numpy()
        return out

    def _process_full_wsi(self, probabilities, *, return_predictions=None):
        """One ``postproc`` over the whole processed region (ref. :999-1076)."""
        outs = self._postproc_maps(probabilities, offset=tuple(int(v) for v in self.mask_padding[:2]))
        flags = [False] * len(outs) if return_predictions is None else list(return_predictions)
        pad_left, pad_top, pad_right, pad_bottom = (int(v) for v in self.mask_padding)
        for task, wanted in zip(outs, flags):
            if not wanted:
                del task["predictions"]
            else:
                task["predictions"] = np.pad(task["predictions"], ((pad_top, pad_bottom), (pad_left, pad_right)))
        return outs

    def _process_tile_mode(self, probabilities, wsi_proc_shape, mask_reader=None, *, return_predictions=None):
        """Post-process tile by tile and merge across the seams (ref. :1078-1287)."""
        rh, rw = probabilities[0].shape[:2]
        tile_sets = self._get_tile_info((rw, rh), wsi_proc_shape, mask_reader)
        tasks = _build_tile_tasks(tile_sets)
        ioconfig = self._ioconfig.to_baseline()
        model = self.model.module if hasattr(self.model, "module") else self.model
        # every tile's post-processing first: tiles of one shape go through the batched device pipeline together
        results: list = [None] * len(tasks)
        by_shape: dict[tuple[int, int], list[int]] = {}
        clipped = []
        for i, (b, _, _) in enumerate(tasks):
            # slicing past the region's end truncates the tile, exactly like the reference's array slicing (:1344-1350)
            x0, y0, x1, y1 = max(int(b[0]), 0), max(int(b[1]), 0), min(int(b[2]), rw), min(int(b[3]), rh)
            clipped.append((x0, y0, x1, y1))
            by_shape.setdefault((y1 - y0, x1 - x0), []).append(i)
        chunk = max(1, int(getattr(self, "tile_batch", 8)))
        engine_dev = torch.device(getattr(self, "device", "cpu"))
        work_dev = engine_dev if (engine_dev.type == "cuda" and not probabilities[0].is_cuda) else None
        # multi-process runs shard the tiles (round-robin inside every shape group); the merge below is sequential and
        # cheap, so every rank repeats it on the gathered tile results and ends with the same tables
        from tiatoolbox_amd import distributed as tdist

        rank, world = tdist.world() if getattr(self, "distributed", True) else (0, 1)
        owned: list[list[int]] = [[] for _ in range(world)]
        for members in by_shape.values():
            for j, i in enumerate(members):
                owned[j % world].append(i)
        for members in by_shape.values():
            members = [i for j, i in enumerate(members) if j % world == rank]
            for s in range(0, len(members), chunk):
                part = members[s:s + chunk]
                crops = [torch.stack([p[clipped[i][1]:clipped[i][3], clipped[i][0]:clipped[i][2]] for i in part])
                         for p in probabilities]
                if work_dev is not None:  # heads streamed to host (slide > HBM): only the tiles in flight live on the device
                    crops = [c.to(work_dev) for c in crops]
                # the batched device pipeline is HoVer-Net's nuclei pass (Sobel-21, 10-pixel objects); models with several
                # tasks (HoVerNet+: Sobel-11 / 3-pixel nuclei at scale 0.5 plus the layer head) go through their own postproc
                if len(getattr(model, "tasks", ())) == 1 and hasattr(model, "postproc_batch"):
                    outs = model.postproc_batch(crops[0], crops[1], crops[2] if len(crops) > 2 else None)  # noqa: PLR2004
                    for j, i in enumerate(part):
                        results[i] = (outs[j],)
                else:
                    for j, i in enumerate(part):
                        results[i] = self._postproc_maps([c[j] for c in crops])
        if world > 1:
            results = self._gather_tile_results(results, [sorted(o) for o in owned], rank, clipped, work_dev or probabilities[0].device,
                                                want_predictions=bool(return_predictions) and any(return_predictions))
        # then the merge, in the reference's tile order
        wsi_info, max_inst = None, None
        for (bounds, flag, mode), out in zip(tasks, results):
            if wsi_info is None:
                flags = [False] * len(out) if return_predictions is None else list(return_predictions)
                wsi_info = tuple({"task_type": t["task_type"],
                                  "predictions": np.zeros(tuple(wsi_proc_shape[::-1]), dtype=t["predictions"].dtype)
                                  if flags[k] else None, "info_dict": {}} for k, t in enumerate(out))
            wsi_info, max_inst = _update_tile_based_predictions_array(out, wsi_info, bounds, self.mask_padding[:2], max_inst)
            tl, br = np.asarray(bounds[:2]), np.asarray(bounds[2:])
            for k, inst_dict in enumerate(_get_inst_info_dicts(out)):
                fresh, stale = _compute_info_dict_for_merge(inst_dict, mode, wsi_info[k]["info_dict"], ioconfig, br - tl, tl, flag)
                wsi_info[k]["info_dict"].update(fresh)
                for key in stale:
                    wsi_info[k]["info_dict"].pop(key, None)
        return self._inst_dict_for_dask_processing(wsi_info)

    def _gather_tile_results(self, results, owned, rank, clipped, device, *, want_predictions: bool):
        """Exchange per-tile post-processing results between ranks (single-task models): instance tables through the
        ragged gather of ``distributed.gather_instance_tables``, tile label maps (only when a slide-sized prediction
        map is requested) as one flat int32 payload split by the tile shapes every rank knows."""
        from tiatoolbox_amd import distributed as tdist

        model = self.model.module if hasattr(self.model, "module") else self.model
        task_type = model.tasks[0]
        mine = owned[rank]
        if any(len(results[i]) != 1 for i in mine) or len(getattr(model, "tasks", ())) > 1:
            # several tasks per tile (HoVerNet+: nuclei + layers, different columns per task): the tiles' small host records
            # (tables with per-task columns) are exchanged as objects, one gather for the whole slide (ref. :1556-1730 keeps one
            # sub-table per task); the tile-sized LABEL MAPS do not go through pickle -- they travel as one flat int32 payload
            # (the ragged gather of the single-task path), described in the records by (shape, dtype) only
            payload, maps = {}, []
            for i in mine:
                recs = []
                for task in results[i]:
                    rec = {k: v for k, v in task.items() if k != "predictions"}
                    rec["predictions"] = None
                    if want_predictions:
                        pred = task["predictions"]
                        pred = pred.cpu().numpy() if isinstance(pred, torch.Tensor) else np.asarray(pred)
                        rec["predictions"] = (tuple(pred.shape), pred.dtype.str)
                        maps.append(pred.astype(np.int32, copy=False).ravel())
                    recs.append(rec)
                payload[i] = tuple(recs)
            parts = tdist.all_gather_objects(payload)
            full = None
            if want_predictions:
                flat = np.concatenate(maps) if maps else np.zeros(0, np.int32)
                full = tdist.all_gather_ragged(torch.from_numpy(np.ascontiguousarray(flat)).to(device))[0].cpu().numpy()
            out: list = [None] * len(results)
            pos = 0
            for part in parts:  # rank order; inside a rank its tiles in the order it packed them = ascending tile index
                for i, tasks_ in part.items():
                    fixed = []
                    for task in tasks_:
                        task = dict(task)
                        if task["predictions"] is None:  # not requested: the merge below never reads it
                            task["predictions"] = np.zeros((0, 0), np.int32)
                        else:
                            shape, dtype = task["predictions"]
                            size = int(np.prod(shape))
                            task["predictions"] = full[pos:pos + size].reshape(shape).astype(np.dtype(dtype), copy=False)
                            pos += size
                        fixed.append(task)
                    out[i] = tuple(fixed)
            return out
        tables = tdist.gather_instance_tables([results[i][0]["info_dict"] for i in mine], device)
        order = [i for tiles in owned for i in tiles]
        preds: dict[int, np.ndarray] = {}
        if want_predictions:
            flat = (np.concatenate([np.asarray(results[i][0]["predictions"], dtype=np.int32).ravel() for i in mine])
                    if mine else np.zeros(0, np.int32))
            full = tdist.all_gather_ragged(torch.from_numpy(np.ascontiguousarray(flat)).to(device))[0].cpu().numpy()
            pos = 0
            for i in order:
                x0, y0, x1, y1 = clipped[i]
                size = (y1 - y0) * (x1 - x0)
                preds[i] = full[pos:pos + size].reshape(y1 - y0, x1 - x0)
                pos += size
        out: list = [None] * len(results)
        for k, i in enumerate(order):
            out[i] = ({"task_type": task_type, "predictions": preds.get(i, np.zeros((0, 0), np.int32)),
                       "info_dict": tables[k], "seg_type": "instance"},)
        return out

    def _post_process_patches_multi_task(self, raw_predictions: dict, model) -> dict:
        """Several tasks per patch (HoVerNet+: nuclei + layers): the model's ``postproc`` per patch on device-resident
        heads; every task gets its own sub-dict ``{predictions, <info columns>}`` (ref. :1556-1685, :1706-1730)."""
        heads = raw_predictions["probabilities"]
        n = heads[0].shape[0]
        per_task: dict[str, list[dict]] = {}
        for i in range(n):
            for task in model.postproc([h[i] for h in heads], offset=(0, 0)):
                per_task.setdefault(task["task_type"], []).append(task)
        out: dict = {}
        for name, items in per_task.items():
            preds = [t["predictions"] for t in items]
            preds = [p.cpu().numpy() if isinstance(p, torch.Tensor) else np.asarray(p) for p in preds]
            sub = {"predictions": np.stack(preds), "seg_type": items[0]["seg_type"]}
            for key in items[0]["info_dict"]:
                sub[key] = [t["info_dict"][key] for t in items]
            out[name] = sub
        if "shard" in raw_predictions:
            # patch-sharded run: every rank post-processed its contiguous shard; the per-task label maps and tables (host
            # records, different columns per task) are gathered in rank order = input order; empty shards contribute nothing
            from tiatoolbox_amd import distributed as tdist

            parts = tdist.all_gather_objects(out)
            names = [name for part in parts for name in part]
            merged: dict = {}
            for name in dict.fromkeys(names):
                have = [part[name] for part in parts if name in part]
                sub = {"predictions": np.concatenate([h["predictions"] for h in have]), "seg_type": have[0]["seg_type"]}
                for key in have[0]:
                    if key not in ("predictions", "seg_type"):
                        sub[key] = [row for h in have for row in h[key]]
                merged[name] = sub
            out = merged
            if self.return_probabilities:
                _, _, n_total = raw_predictions["shard"]
                heads = [tdist.all_gather_rows(h if isinstance(h, torch.Tensor) else torch.from_numpy(np.asarray(h)), n_total)
                         for h in heads]
        self.tasks = set(out) or set(model.tasks)
        if self.return_probabilities:
            out["probabilities"] = [h.cpu().numpy() if isinstance(h, torch.Tensor) else h for h in heads]
        return out

    def _inst_dict_for_dask_processing(self, wsi_info_dict, keys_to_shift=("centroid", "box", "contours")):
        """Row-wise instance records -> one object array per key, shifted back into slide coordinates
        (ref. :1289-1329; NumPy arrays instead of dask arrays)."""
        offset = np.array(self.mask_padding[:2])
        for task in wsi_info_dict or ():
            records = list(task["info_dict"].values())
            cols = {}
            for key in (records[0] if records else {}):
                col = np.empty(len(records), dtype=object)
                for i, rec in enumerate(records):
                    col[i] = rec[key]
                cols[key] = apply_coordinate_offset(col, offset, key, keys_to_shift)
            task["info_dict"] = cols
        return wsi_info_dict

    def post_process_wsi(self, raw_predictions: dict, wsi_proc_shape, mask_reader=None, *, return_predictions=None) -> dict:
        """Full-region or tile-mode post-processing, organised per task (ref. :836-997)."""
        probabilities = raw_predictions["probabilities"]
        tile_h, tile_w = self._ioconfig.tile_shape
        if any(p.shape[0] > tile_h or p.shape[1] > tile_w for p in probabilities):
            outs = self._process_tile_mode(probabilities, wsi_proc_shape, mask_reader, return_predictions=return_predictions)
        else:
            outs = self._process_full_wsi(probabilities, return_predictions=return_predictions)
        self.tasks = set()
        for task in outs or ():
            name = task["task_type"]
            self.tasks.add(name)
            raw_predictions[name] = {}
            for key, value in task.items():
                if key == "task_type":
                    continue
                if isinstance(value, np.ndarray):
                    raw_predictions[name][key] = value
                elif isinstance(value, dict):
                    raw_predictions[name].update(value)
        return raw_predictions

    def run(self, images, *, masks=None, patch_mode: bool = True, ioconfig=None, return_predictions=None, save_dir=None,
            overwrite: bool = False, output_type: str = "dict", **kwargs):
        """Patch mode: ``EngineABC.run``.  WSI mode (ref. ``engine_abc.py:1684-1829``, ``multi_task_segmentor.py:2088-2300``):
        ``images`` = list of ``ArrayWSIReader`` / HxWx3 arrays / ``.npy`` paths, ``save_dir`` REQUIRED (``OSError`` otherwise;
        created with the reference's ``overwrite`` rule); per slide one ``<stem>.npz`` holding the task's instance table
        (``box`` / ``centroid`` / ``contours`` / ``prob`` / ``type``; object arrays: ``np.load(..., allow_pickle=True)``), the
        patch ``coordinates`` and, on request, ``predictions`` / ``probabilities``; returns ``{image key: Path}``.
        :meth:`process_wsi` is the in-memory form of one slide."""
        if patch_mode:
            return super().run(images, masks=masks, patch_mode=True, ioconfig=ioconfig, save_dir=save_dir, overwrite=overwrite,
                               output_type=output_type, **kwargs)
        if kwargs.get("return_labels"):
            msg = "`return_labels` is not supported for MultiTaskSegmentor."
            raise ValueError(msg)  # ref. :2149-2156
        self._update_run_params(images=images, masks=masks, save_dir=save_dir, ioconfig=ioconfig, output_type=output_type,
                                overwrite=overwrite, patch_mode=False, **kwargs)
        from pathlib import Path

        from tiatoolbox_amd import distributed as tdist
        from tiatoolbox_amd.models.engine.engine_abc import prepare_engines_save_dir, write_outputs

        save_dir = prepare_engines_save_dir(save_dir, patch_mode=False, overwrite=overwrite, distributed=self.distributed)
        paths: dict = {}
        for i, image in enumerate(self.images):
            mask = self.masks[i] if self.masks is not None else None
            out = self.process_wsi(image, mask, return_predictions=return_predictions,
                                   auto_get_mask=kwargs.get("auto_get_mask", True))
            key = image if isinstance(image, (str, Path)) else i
            stem = Path(image).stem if isinstance(image, (str, Path)) else str(i)
            paths[key] = save_dir / f"{stem}.npz"

            def write(out=out, path=paths[key]) -> None:
                flat = {}
                for name, val in out.items():
                    if isinstance(val, dict):  # several tasks: one sub-table each
                        flat.update({f"{name}/{sub}": np.asarray(v) for sub, v in val.items()})
                    elif isinstance(val, list):
                        flat.update({f"{name}/{j}": np.asarray(v) for j, v in enumerate(val)})
                    else:
                        flat[name] = np.asarray(val)
                np.savez(path, **flat)

            write_outputs(self.distributed, write)
        return paths

    def process_wsi(self, image, mask=None, *, return_predictions=None, auto_get_mask: bool = True) -> dict:
        """One slide in memory: tissue mask -> patch inference stitched per head -> full-region or tile-mode post-processing;
        the dict ``run`` writes for it (single task: the task's table at the top level, ref. :1695-1704)."""
        reader = self._open_slide(image)
        mask_reader = None
        if mask is not None:
            mask_reader = self._open_slide(mask, as_mask=True)
        elif auto_get_mask:
            mask_reader = reader.tissue_mask(resolution=1.25, units="power")
        raw = self.infer_wsi(reader, mask_reader)
        if raw["probabilities"] is None:
            return {"coordinates": raw["coordinates"]}
        out = self.post_process_wsi(raw, reader.slide_dimensions, mask_reader, return_predictions=return_predictions)
        heads = out.pop("probabilities")
        if self.return_probabilities:
            pad_left, pad_top, pad_right, pad_bottom = (int(v) for v in self.mask_padding)
            out["probabilities"] = [np.pad(p.cpu().numpy(), ((pad_top, pad_bottom), (pad_left, pad_right), (0, 0)))
                                    for p in heads]
        if len(self.tasks) == 1:  # single task: its table moves to the top level (ref. :1695-1704)
            out.update(out.pop(next(iter(self.tasks))))
            out.pop("seg_type", None)
        return out

    predict = run  # tiatoolbox 1.x name

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
