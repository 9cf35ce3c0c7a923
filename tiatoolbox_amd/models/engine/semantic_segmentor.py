"""Semantic segmentation engine (API of reference ``tiatoolbox/models/engine/semantic_segmentor.py``).

Patch mode = PatchPredictor with dense outputs.  WSI mode stitches overlapping patch outputs into a
whole-slide prediction entirely on the GPU: per patch row a gather kernel merges the row
(``tia_canvas_row_merge_f32``), consecutive rows are added over their overlap, normalised by the
count and arg-maxed (``tia_canvas_finalize_f32``) -- the reference's ``merge_horizontal`` /
``merge_vertical_chunkwise`` arithmetic, bit for bit, without the 5 MiB/patch device->host copies.
Patch rows are sharded over ranks when ``torch.distributed`` is initialised.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from tiatoolbox_amd.utils import tracing

from tiatoolbox_amd import _lib, distributed as tdist
from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor
from tiatoolbox_amd.tools.patchextraction import PatchExtractor
from tiatoolbox_amd.wsicore import ArrayWSIReader


def merge_batch_to_canvas(blocks, output_locations, merged_shape):
    """Row merge of the reference (:1141-1183) on the GPU; NumPy in/out for API compatibility.

    Blocks must share one shape and one ``ys`` (a patch row), as in every call the reference makes.
    """
    blocks_t = torch.as_tensor(np.asarray(blocks, dtype=np.float32))
    locs = np.asarray(output_locations).reshape(-1, 4).astype(np.int64)
    h, w, c = merged_shape
    from tiatoolbox_amd.utils._tensors import default_device

    dev = default_device()
    if blocks_t.shape[0] == 0:
        return np.zeros(merged_shape, dtype=np.asarray(blocks).dtype), np.zeros((h, w, 1), dtype=np.uint8)
    order = np.argsort(locs[:, 0], kind="stable")
    row, cnt = _row_merge(blocks_t[order].to(dev), locs[order, 0], w)
    canvas = row[:h].cpu().numpy().astype(np.asarray(blocks).dtype)
    return canvas, cnt[:h].cpu().numpy()[..., None]


def _row_merge(blocks: torch.Tensor, xs, width: int) -> tuple[torch.Tensor, torch.Tensor]:
    """``xs``: the patches' left edges -- a NumPy array, or an int32 tensor already on the device (no copy, no sync)."""
    n, oh, ow, c = blocks.shape
    blocks = blocks.contiguous()
    row = torch.empty((oh, width, c), dtype=torch.float32, device=blocks.device)
    cnt = torch.empty((oh, width), dtype=torch.uint8, device=blocks.device)
    flags = torch.empty(n, dtype=torch.int32, device=blocks.device)
    if isinstance(xs, torch.Tensor) and xs.is_cuda and xs.dtype == torch.int32:
        xs_t = xs.contiguous()
    else:
        xs_t = torch.as_tensor(np.asarray(xs, dtype=np.int32)).to(blocks.device)
    with torch.cuda.device(blocks.device):
        rc = _lib.load().tia_canvas_row_merge_f32(blocks.data_ptr(), xs_t.data_ptr(), n, oh, ow, c, width, row.data_ptr(),
                                                  cnt.data_ptr(), flags.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_canvas_row_merge_f32")
    return row, cnt


def _finalize(row_a, cnt_a, ys_a, row_b, cnt_b, ys_b, y0, y1, probs, pred, *, y_base: int = 0) -> None:
    """Normalise + arg-max canvas rows ``[y0, y1)``.  ``pred`` / ``probs`` may be a band of the slide-sized maps whose
    first row is slide row ``y_base``: row coordinates are shifted by ``-y_base`` (the kernel only ever uses the
    differences ``y - ys``)."""
    oh, width, c = row_a.shape
    with torch.cuda.device(row_a.device):
        rc = _lib.load().tia_canvas_finalize_f32(
            row_a.data_ptr(), cnt_a.data_ptr(), int(ys_a) - y_base, row_b.data_ptr() if row_b is not None else 0,
            cnt_b.data_ptr() if cnt_b is not None else 0, int(ys_b) - y_base, oh, width, c, int(y0) - y_base, int(y1) - y_base,
            probs.data_ptr() if probs is not None else 0, pred.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_canvas_finalize_f32")


def band_plan(row_ys: np.ndarray, oh: int, height: int, rank: int, world: int) -> dict:
    """Which patch rows a rank infers and which canvas rows it owns (contiguous bands of patch rows).

    ``rows`` = indices into ``row_ys`` the rank must infer (its own rows plus one leading row, whose lower part overlaps
    the band's first canvas rows); ``own`` = the slice of ``rows`` it finalises; ``y_lo:y_hi`` = canvas rows it owns;
    ``bands`` = ``(y_lo, y_hi)`` of every rank, so all ranks agree on the layout of the exchange."""
    n_rows = len(row_ys)

    def limits(r: int) -> tuple[int, int, int, int]:
        lo, hi = tdist.shard_bounds(n_rows, r, world)
        y_lo = min(int(row_ys[lo]), height) if lo < n_rows else height
        y_hi = min(int(row_ys[hi]), height) if hi < n_rows else height
        if lo >= hi:
            y_lo = y_hi = height
        return lo, hi, y_lo, y_hi

    lo, hi, y_lo, y_hi = limits(rank)
    # a rank without rows of its own (more ranks than patch rows) infers nothing: no leading row either
    return {"rows": list(range(max(lo - 1, 0), hi)) if lo < hi else [], "own": (lo, hi), "y_lo": y_lo, "y_hi": y_hi,
            "bands": [limits(r)[2:] for r in range(world)], "oh": oh}


def exchange_bands(local: torch.Tensor, plan: dict, height: int) -> torch.Tensor:
    """All-gather of the rank-local canvas bands into the full map (every rank gets it): one
    ``all_gather_into_tensor`` of bands padded to the tallest one -- each rank contributes only the rows it owns, so
    the exchange moves the map once instead of all-reducing a mostly-zero full-size copy per rank.

    ``local`` = this rank's band ``[y_hi - y_lo, W, ...]``; works on CPU tensors too (gloo), which is how the
    collective logic is tested without a GPU."""
    bands = plan["bands"]
    world = len(bands)
    if world == 1:
        return local
    tallest = max(max(b[1] - b[0] for b in bands), 1)
    pad = torch.zeros((tallest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * tallest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    torch.distributed.all_gather_into_tensor(out, pad.contiguous())
    full = torch.zeros((height, *local.shape[1:]), dtype=local.dtype, device=local.device)
    for r, (y_lo, y_hi) in enumerate(bands):
        if y_hi > y_lo:
            full[y_lo:y_hi] = out[r * tallest: r * tallest + (y_hi - y_lo)]
    return full


class CanvasBand:
    """Where a rank's finalised canvas rows go ("device canvas band sized to HBM; spill to host pinned memory";
    the reference spills its canvas to zarr above ``memory_threshold``, ``semantic_segmentor.py:552-583,1693-1730``).

    *resident* (a slide whose maps fit the device): one device tensor ``[band_h, W, ...]`` per map, ``_finalize`` writes into it
    at its slide row -- the round-1..4 behaviour.  *streamed* (``device_rows`` = K >= 2): the device holds a ring of K chunks of
    ``oh`` canvas rows per map; a patch row's finished rows ``[ys, y1)`` are finalised into the next chunk and copied to a
    page-locked HOST map on a copy stream while the next patch row is inferred; a chunk is reused only after its copy has
    finished.  Peak device memory = K chunks + the two patch rows being merged, whatever the slide's height.  The maps come back
    as host tensors; the bytes are the resident mode's, bit for bit (same kernel, same arguments up to the row offset).

    ``maps``: ``{name: (trailing shape, dtype)}``, e.g. ``{"pred": ((), uint8), "probs": ((5,), float32)}``."""

    def __init__(self, band_h: int, width: int, y_lo: int, oh: int, device: torch.device, maps: dict, *,
                 device_rows: int | None = None) -> None:
        self.band_h, self.width, self.y_lo, self.oh, self.device = int(band_h), int(width), int(y_lo), int(oh), device
        self.streamed = device_rows is not None
        self.maps = dict(maps)
        self._slot = 0
        if not self.streamed:
            self.full = {k: torch.zeros((self.band_h, self.width, *shape), dtype=dt, device=device) for k, (shape, dt) in maps.items()}
            return
        self.k = max(2, int(device_rows))
        self.chunks = [{k: torch.zeros((self.oh, self.width, *shape), dtype=dt, device=device) for k, (shape, dt) in maps.items()}
                       for _ in range(self.k)]
        self.free = [None] * self.k  # per chunk: the event after which its last copy has left the device
        self.copy_stream = torch.cuda.Stream(device=device)
        self.full = {k: self._host((self.band_h, self.width, *shape), dt) for k, (shape, dt) in maps.items()}

    @staticmethod
    def _host(shape, dtype) -> torch.Tensor:
        try:
            return torch.zeros(shape, dtype=dtype, pin_memory=True)
        except RuntimeError:  # more than the driver will page-lock: pageable memory (the copies then block the copy stream's host side only)
            return torch.zeros(shape, dtype=dtype)

    @staticmethod
    def bytes_needed(band_h: int, width: int, maps: dict) -> int:
        return sum(int(band_h) * int(width) * int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size()
                   for shape, dt in maps.values())

    def target(self, ys: int) -> tuple[dict, int]:
        """Tensors to finalise rows starting at slide row ``ys`` into, and the slide row their row 0 stands for (``y_base``)."""
        if not self.streamed:
            return self.full, self.y_lo
        ev = self.free[self._slot]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return self.chunks[self._slot], int(ys)

    def done(self, ys: int, y1: int) -> None:
        """Rows ``[ys, y1)`` are final: streamed mode hands them to the copy stream and moves to the next chunk."""
        if not self.streamed or y1 <= ys:
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        chunk = self.chunks[self._slot]
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            for k, host in self.full.items():
                host[ys - self.y_lo:y1 - self.y_lo].copy_(chunk[k][:y1 - ys], non_blocking=True)
            gone = torch.cuda.Event()
            gone.record(self.copy_stream)
        self.free[self._slot] = gone
        self._slot = (self._slot + 1) % self.k

    def result(self) -> dict:
        if self.streamed:
            self.copy_stream.synchronize()
        return self.full


def _band_device_rows(engine, nbytes: int, device: torch.device, *, exchange_bytes: int = 0, world: int = 1) -> int | None:
    """``None`` = keep the band's maps resident on the device; K = stream them through a ring of K chunks (``CanvasBand``).
    ``engine.device_band_rows`` forces K; otherwise streaming starts when the maps -- plus, with several ranks, what the resident
    band exchange allocates on the device (``exchange_bytes``: the padded band, the gathered bands and the full-height map) --
    would take more than ``memory_threshold`` per cent (the reference's run kwarg, default 80) of the device memory that is free
    right now.  With ``world > 1`` the ranks AGREE on the answer (one MAX all-reduce: any rank that must stream makes all of them
    stream, with the largest K): free memory and band heights differ between ranks, and a resident and a streamed rank would issue
    different collective sequences in the exchange (one all-gather of the tallest band vs one per 2048 rows).  Every rank calls
    this exactly once per slide."""
    forced = getattr(engine, "device_band_rows", None)
    k = max(2, int(forced)) if forced is not None else 4
    stream = forced is not None
    if not stream:
        free = torch.cuda.mem_get_info(device)[0] if device.type == "cuda" else 1 << 62
        threshold = float(getattr(engine, "memory_threshold", 80) or 80)
        stream = nbytes + exchange_bytes > free * threshold / 100.0
    if world > 1:
        from tiatoolbox_amd import distributed as tdist

        flag, k = tdist.agree_max([int(stream), k if stream else 0], device)
        stream = bool(flag)
        k = max(k, 2)
    return k if stream else None


def exchange_footprint(plan: dict, height: int, width: int, maps: dict) -> int:
    """Device bytes of the resident ``exchange_bands`` of ``maps`` (one after the other, the largest counts): the padded band, the
    ``world`` gathered bands and the full-height result -- rank-independent quantities."""
    bands = plan["bands"]
    world = len(bands)
    if world == 1:
        return 0
    tallest = max(max(b[1] - b[0] for b in bands), 1)
    per_row = max(int(width) * int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dt).element_size() for shape, dt in maps.values())
    return per_row * (tallest + world * tallest + int(height))


def exchange_bands_streamed(local: torch.Tensor, plan: dict, height: int, device: torch.device, rows: int = 2048) -> torch.Tensor:
    """``exchange_bands`` for HOST bands of a slide that does not fit the device: the same padded all-gather, ``rows`` canvas rows at
    a time through the device (the collective's tensors must live there under RCCL), every rank assembling the full HOST map."""
    bands = plan["bands"]
    world = len(bands)
    if world == 1:
        return local
    tallest = max(max(b[1] - b[0] for b in bands), 1)
    full = torch.zeros((height, *local.shape[1:]), dtype=local.dtype)
    on_dev = device.type == "cuda"
    for r0 in range(0, tallest, rows):
        n = min(rows, tallest - r0)
        pad = torch.zeros((n, *local.shape[1:]), dtype=local.dtype, device=device)
        mine = local[r0:r0 + n]
        if mine.shape[0]:
            pad[: mine.shape[0]] = mine.to(device, non_blocking=False)
        out = torch.empty((world * n, *local.shape[1:]), dtype=local.dtype, device=device)
        torch.distributed.all_gather_into_tensor(out, pad.contiguous())
        out_h = out.cpu() if on_dev else out
        for r, (y_lo, y_hi) in enumerate(bands):
            take = min(max(y_hi - y_lo - r0, 0), n)
            if take > 0:
                full[y_lo + r0:y_lo + r0 + take] = out_h[r * n:r * n + take]
    return full


class SemanticSegmentor(PatchPredictor):
    """Semantic segmentation of patches or whole (in-memory) slides (ref. :136-1821)."""

    def __init__(self, model, batch_size: int = 8, num_workers: int = 0, weights=None, *, device: str = "cpu",
                 verbose: bool = True) -> None:
        super().__init__(model=model, batch_size=batch_size, num_workers=num_workers, weights=weights, device=device,
                         verbose=verbose)
        self.auto_get_mask = True
        self.fold_batchnorm = False  # the decoder is pre-activation (BN before conv): keep the module as is

    # ------------------------------------------------------------------------------ WSI mode
    def get_coordinates(self, reader: ArrayWSIReader, mask_reader: ArrayWSIReader | None):
        cfg = self._ioconfig
        w, h = reader.slide_dimensions
        in_b, out_b = PatchExtractor.get_coordinates(
            patch_output_shape=tuple(cfg.patch_output_shape[::-1]), image_shape=(w, h),
            patch_input_shape=tuple(cfg.patch_input_shape[::-1]), stride_shape=tuple(cfg.stride_shape[::-1]))
        keep = np.ones(len(in_b), dtype=bool)
        if mask_reader is not None:
            keep = PatchExtractor.filter_coordinates(mask_reader, out_b, (w, h), min_mask_ratio=0)
        return in_b, out_b, keep

    def infer_wsi(self, reader: ArrayWSIReader, mask_reader: ArrayWSIReader | None = None, *,
                  return_probabilities: bool = False) -> dict:
        """Tile, infer and stitch one slide; returns device tensors -- or, for a slide whose maps would not fit the device
        (``memory_threshold``, or ``self.device_band_rows`` set), HOST tensors streamed there band by band (``CanvasBand``)."""
        dev = torch.device(self.device)
        if dev.type != "cuda":
            msg = "WSI-mode stitching runs on the GPU (device='cuda'); there is no CPU fallback."
            raise _lib.HipLibraryError(msg)
        from tiatoolbox_amd.models.engine.engine_abc import _DTYPES

        dtype = _DTYPES[str(self.compute_dtype).replace("torch.", "")]
        model = self._inference_model(dtype)
        infer_batch = self._get_model_attr("infer_batch")
        w, h = reader.slide_dimensions
        in_b, out_b, keep = self.get_coordinates(reader, mask_reader)
        row_ys = np.unique(out_b[:, 1])
        oh = int(self._ioconfig.patch_output_shape[0])
        rank, world = tdist.world() if self.distributed else (0, 1)
        plan = band_plan(row_ys, oh, h, rank, world)
        r_lo, r_hi = plan["own"]
        y_lo, y_hi = plan["y_lo"], plan["y_hi"]
        # rank-local band of the slide-sized maps (row 0 of the band = canvas row y_lo): resident on the device, or -- a slide
        # whose maps do not fit -- streamed to page-locked host memory patch row by patch row (CanvasBand)
        band_h = max(y_hi - y_lo, 0)
        band = None
        n_ch = None
        prev = None  # (row, cnt, ys)

        def open_band(channels: int) -> CanvasBand:
            maps = {"pred": ((), torch.uint8)}
            if return_probabilities:
                maps["probs"] = ((channels,), torch.float32)
            gathers = world > 1 and not getattr(self, "return_bands", False)
            k = _band_device_rows(self, CanvasBand.bytes_needed(band_h, w, maps), dev, world=world,
                                  exchange_bytes=exchange_footprint(plan, h, w, maps) if gathers else 0)
            return CanvasBand(band_h, w, y_lo, oh, dev, maps, device_rows=k)

        from tiatoolbox_amd.models.engine.engine_abc import iter_row_outputs

        rows = list(plan["rows"])
        row_sels = [np.flatnonzero((out_b[:, 1] == int(row_ys[ri])) & keep) for ri in rows]
        # Every patch's input bounds and output x offset go to the device ONCE, in the order the loop consumes them
        # (`iter_row_outputs`: batches of `batch_size` over the concatenated rows, the last one padded with its last patch):
        # a per-batch upload is a synchronous copy that makes the host wait for the previous batch's kernels, so the next
        # forward's launches would not overlap them (host profile: 7 % of a 20,000^2 slide).
        flat = np.concatenate([s for s in row_sels if len(s)]) if any(len(s) for s in row_sels) else np.zeros(0, np.int64)
        bs = int(self.batch_size)
        size = (int(in_b[0, 2] - in_b[0, 0]), int(in_b[0, 3] - in_b[0, 1])) if len(in_b) else (0, 0)
        uniform = len(in_b) > 0 and bool(np.all(in_b[:, 2] - in_b[:, 0] == size[0]) and np.all(in_b[:, 3] - in_b[:, 1] == size[1]))
        bounds_dev = xs_dev = None
        if uniform and len(flat):
            padded = np.concatenate([flat, np.repeat(flat[-1:], bs)])
            bounds_dev = torch.from_numpy(np.ascontiguousarray(in_b[padded], dtype=np.int32)).to(dev)
            xs_dev = torch.from_numpy(np.ascontiguousarray(out_b[flat, 0], dtype=np.int32)).to(dev)
        row_starts = np.concatenate([[0], np.cumsum([len(s) for s in row_sels])])
        state = {"pos": 0}

        def infer(idx):
            if bounds_dev is None:
                return infer_batch(model, reader.read_bounds_batch(in_b[idx]), device=self.device)
            p = state["pos"]
            state["pos"] = p + min(bs, len(flat) - p)  # the batches walk `flat` front to back (see iter_row_outputs)
            return infer_batch(model, reader.read_bounds_batch(bounds_dev[p:p + bs], size=size), device=self.device)

        with self._miopen_scope():
            for k, blocks in iter_row_outputs(infer, row_sels, self.batch_size):
                ri, sel = rows[k], row_sels[k]
                ys = int(row_ys[ri])
                if blocks is not None:
                    n_ch = blocks.shape[-1]
                    xs = xs_dev[int(row_starts[k]):int(row_starts[k + 1])] if xs_dev is not None else out_b[sel, 0]
                    with tracing.range("canvas_row_merge"):
                        row, cnt = _row_merge(blocks, xs, w)
                else:
                    if n_ch is None:
                        probe = infer_batch(model, reader.read_bounds_batch(in_b[:1]), device=self.device)
                        n_ch = probe.shape[-1]
                    row = torch.zeros((oh, w, n_ch), dtype=torch.float32, device=dev)
                    cnt = torch.zeros((oh, w), dtype=torch.uint8, device=dev)
                if band is None:
                    band = open_band(n_ch)
                if ri >= r_lo:
                    y1 = min(int(row_ys[ri + 1]) if ri + 1 < len(row_ys) else ys + oh, h)
                    dst, y_base = band.target(ys)
                    if prev is None:
                        _finalize(row, cnt, ys, None, None, 0, ys, y1, dst.get("probs"), dst["pred"], y_base=y_base)
                    else:
                        _finalize(prev[0], prev[1], prev[2], row, cnt, ys, ys, y1, dst.get("probs"), dst["pred"], y_base=y_base)
                    band.done(ys, y1)
                prev = (row, cnt, ys)
        if band is None:
            band = open_band(n_ch or 1)
        maps = band.result()
        pred, probs = maps["pred"], maps.get("probs")
        self.last_band_streamed = band.streamed
        if world > 1 and not getattr(self, "return_bands", False):
            gather = (lambda t: exchange_bands_streamed(t, plan, h, dev)) if band.streamed else (lambda t: exchange_bands(t, plan, h))
            pred = gather(pred)
            if return_probabilities:
                probs = gather(probs)
        out = {"predictions": pred, "coordinates": out_b[keep], "band": (y_lo, y_hi)}
        if return_probabilities and probs is not None:
            out["probabilities"] = probs
        return out

    def run(self, images, *, masks=None, input_resolutions=None, patch_input_shape=None, ioconfig=None,
            patch_mode: bool = True, save_dir=None, overwrite: bool = False, output_type: str = "dict", **kwargs):
        """Patch mode: as PatchPredictor.  WSI mode (ref. ``engine_abc.py:1684-1829`` + ``semantic_segmentor.py:393-631``):
        ``images`` is a list of ``ArrayWSIReader`` / HxWx3 arrays / ``.npy`` paths and ``save_dir`` is REQUIRED
        (``OSError`` otherwise, created with the reference's ``overwrite`` rule); per slide one ``<stem>.npz`` with
        ``predictions`` (HxW uint8), ``coordinates`` and, with ``return_probabilities=True``, ``probabilities``
        (HxWxC float32) -- the members of the reference's zarr store; returns ``{image key: Path}``.
        :meth:`infer_wsi` is the in-memory form (device tensors, nothing written)."""
        if patch_mode:
            return super().run(images, masks=masks, input_resolutions=input_resolutions,
                               patch_input_shape=patch_input_shape, ioconfig=ioconfig, patch_mode=True,
                               save_dir=save_dir, overwrite=overwrite, output_type=output_type, **kwargs)
        self._update_run_params(images=images, masks=masks, input_resolutions=input_resolutions,
                                patch_input_shape=patch_input_shape, save_dir=save_dir, ioconfig=ioconfig,
                                output_type=output_type, overwrite=overwrite, patch_mode=False, **kwargs)
        from tiatoolbox_amd.models.engine.engine_abc import prepare_engines_save_dir, write_outputs

        save_dir = prepare_engines_save_dir(save_dir, patch_mode=False, overwrite=overwrite, distributed=self.distributed)
        paths: dict = {}
        for i, image in enumerate(self.images):
            reader = self._open_slide(image)
            mask_reader = None
            if self.masks is not None:
                mask_reader = self._open_slide(self.masks[i], as_mask=True)
            elif self.auto_get_mask:
                mask_reader = reader.tissue_mask(resolution=1.25, units="power")
            out = self.infer_wsi(reader, mask_reader, return_probabilities=bool(self.return_probabilities))
            arrays = {"predictions": out["predictions"].cpu().numpy(), "coordinates": out["coordinates"]}
            if "probabilities" in out:
                arrays["probabilities"] = out["probabilities"].cpu().numpy()
            key = image if isinstance(image, (str, Path)) else i
            stem = Path(image).stem if isinstance(image, (str, Path)) else str(i)
            paths[key] = save_dir / f"{stem}.npz"
            write_outputs(self.distributed, lambda path=paths[key], arrays=arrays: np.savez(path, **arrays))
        return paths

    predict = run
