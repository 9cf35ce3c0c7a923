"""``bench.py --config semantic | hovernet | vahadane``: BASELINE.json configs[2], configs[3], configs[4].

Same contract as the headline bench (one JSON line on rank 0, W untimed warm-up steps, K timed steps between
barrier + synchronize, max over ranks, ``roofline`` from HIP events on the launch stream, ``cpu_baseline`` = the oracle
on a bounded sample on rank 0 at N=1).  A *step* is one pass of the config's whole job over the rank's share:

* ``semantic`` -- ``SemanticSegmentor("fcn_resnet50_unet-bcss").run([slide], patch_mode=False)`` over ONE synthetic
  ``--slide`` x ``--slide`` (default 20 000) in-memory WSI: Otsu tissue mask of the thumbnail, patch grid + mask filter,
  device patch gather, UNet-R50 fp32, device stitching; patch rows sharded over ranks, bands all-gathered (strong scaling:
  the slide is the job).  Unit: 1024x1024 input patches/s.
* ``hovernet`` -- ``NucleusInstanceSegmentor("hovernet_fast-pannuke").run(patches, patch_mode=True)`` on 256x256 tiles:
  HoVer-Net fast fp32 + the HIP post-processing (Sobel-21 ... watershed, instance statistics, contours).  Unit: tiles/s.
  The post-processing is also timed alone on synthetic head outputs with ~60 nuclei per tile.
* ``vahadane`` -- ``VahadaneNormalizer.transform`` (dictionary learning on the device + normalise) followed by
  ``StainAugmentor(method="vahadane").fit + augment`` over 256x256 patches (8192 per GPU = 65 536 over 8).  Unit: patches/s.
"""

from __future__ import annotations

import os
import statistics
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
HBM_PEAK_GBS = 8000.0
MFMA_PEAK_F32 = 157.3


_RCCL: dict | None = None  # what bench.verify_ranks learned about the process group (N > 1): reported as `rccl` in every line


def _setup(args):
    import logging

    import torch

    from tiatoolbox_amd import distributed as tdist

    rank, world_size, local_rank = tdist.init_from_env()
    torch.cuda.set_device(local_rank)
    import bench

    global _RCCL  # noqa: PLW0603
    _RCCL = bench.verify_ranks(args, world_size, local_rank)
    logging.getLogger("tiatoolbox_amd").setLevel(logging.ERROR)
    return rank, world_size, torch.device("cuda", local_rank)


def _traffic(stem: str, kernel_substrs: tuple, per_call_kernel: str) -> dict:
    """``traffic`` (HBM-side bytes per call of the timed operation) from this round's committed ``--pmc`` passes over the same
    shapes (``scripts/pmc_workloads.py``), or ``None`` when no pass is committed."""
    import bench

    got = bench.pmc_traffic_per_call(stem, kernel_substrs, per_call_kernel)
    if got is None:
        return {"traffic": None}
    return {"traffic": int(got["bytes"]), "traffic_source": got["source"]}


def _vahadane_cpu_worker(job):
    """Pool worker of the all-core Vahadane CPU baseline: oracle transform + augmentor fit / augment of one patch."""
    import numpy as np

    from oracle import stain as ostain

    target, patch = job
    global _VAHADANE_REF  # noqa: PLW0603  (one fitted normaliser per worker process)
    try:
        ref = _VAHADANE_REF
    except NameError:
        ref = ostain.get_normalizer("vahadane")
        ref.fit(target.copy())
        _VAHADANE_REF = ref
    nrm = ref.transform(patch.copy())
    sm = ostain.VahadaneExtractor(random_state=0).get_stain_matrix(nrm.copy())
    ostain.stain_augment(nrm, sm, np.array([1.1, 0.9]), np.array([0.05, -0.05]), threshold=0.85)
    return 0


def _timed(step, args, world_size: int, device) -> float:
    """W untimed + K timed steps between barrier + synchronize; returns the MAX over ranks of the timed region.  Each rank's
    own time (before it waits for the others at the closing barrier) is kept in ``_timed.own`` for ``_per_rank``."""
    import torch

    def sync() -> None:
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize()

    def barrier() -> None:
        if world_size > 1:
            torch.distributed.barrier()
        sync()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    _timed.own = (time.perf_counter() - t0) / max(args.steps, 1) * 1e3
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    if world_size > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


_timed.own = 0.0


def _per_rank(world_size: int, device, collective=None, *, what: str, nbytes: int = 0, reps: int = 10) -> dict | None:
    """The block a scaling run needs to explain itself (same shape as the headline's ``per_rank``): every rank's own step
    time, and the config's collective alone -- ``collective()`` timed over ``reps`` back-to-back calls between barriers
    (``None``: the config has no collective on its data path).  COLLECTIVE on every rank; returns ``None`` at N = 1."""
    import torch

    if world_size <= 1:
        return None
    rank = torch.distributed.get_rank()
    owns = torch.zeros(world_size, dtype=torch.float64, device=device)
    owns[rank] = _timed.own
    torch.distributed.all_reduce(owns)
    out = {"own_ms_per_step": [round(float(v), 3) for v in owns.cpu().tolist()]}
    if collective is not None:
        collective()
        cuda = torch.device(device).type == "cuda"
        torch.distributed.barrier()
        if cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            collective()
        if cuda:
            torch.cuda.synchronize()
        torch.distributed.barrier()
        out["collective_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        out["collective_bytes_per_rank"] = int(nbytes)
    out["what"] = ("own = rank-local wall time per step before the closing barrier (the step already contains its collectives, "
                   "which synchronise the ranks); collective = " + what)
    return out


def _ev(fn, reps: int = 5) -> float:
    import torch

    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def _median_time(fn, reps: int = 3) -> float:
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def _flops_of(model, x) -> float:
    """FLOPs of one forward of the plain torch module on an input of x's shape, counted on the ``meta`` device: shapes only, no
    kernel runs (a real forward of the plain module would go through MIOpen and put its kernels into the traces of this bench)."""
    import copy

    import torch
    from torch.utils.flop_counter import FlopCounterMode

    shadow = copy.deepcopy(model).to("meta").eval()
    with torch.inference_mode(), FlopCounterMode(display=False) as fc:
        shadow(torch.empty(tuple(x.shape), dtype=torch.float32, device="meta"))
    return float(fc.get_total_flops())


def semantic_per_rank(out_b, oh: int, side: int, rank: int, world_size: int, device) -> dict | None:
    """``per_rank`` of the semantic config: its one collective alone -- the all-gather of the rank-local uint8 prediction bands
    of a ``side``-wide slide whose patch outputs are at ``out_b``.  Works on CPU tensors under gloo (tests/test_distributed.py)."""
    import numpy as np
    import torch

    if world_size <= 1:
        return None
    from tiatoolbox_amd.models.engine.semantic_segmentor import band_plan, exchange_bands

    plan = band_plan(np.unique(np.asarray(out_b)[:, 1]), oh, side, rank, world_size)
    band = torch.zeros((max(plan["y_hi"] - plan["y_lo"], 0), side), dtype=torch.uint8, device=device)
    tallest = max(max(b[1] - b[0] for b in plan["bands"]), 1)
    return _per_rank(world_size, device, lambda: exchange_bands(band, plan, side), nbytes=tallest * side,
                     what="exchange_bands: one padded all_gather_into_tensor of the rank-local uint8 prediction bands")


def hovernet_per_rank(out: dict, rank: int, world_size: int, device) -> dict | None:
    """``per_rank`` of the instance config: its collectives alone, on this rank's shard of the finished result ``out`` -- the padded
    all-gather of the int32 label maps and the ragged gathers of the instance tables.  CPU tensors under gloo work too."""
    import numpy as np
    import torch

    if world_size <= 1:
        return None
    from tiatoolbox_amd import distributed as tdist

    n_total = int(out["predictions"].shape[0])
    lo, hi = tdist.shard_bounds(n_total, rank, world_size)
    maps = torch.from_numpy(np.ascontiguousarray(np.asarray(out["predictions"][lo:hi]).astype(np.int32))).to(device)
    tables = [{k: out[k][i] for k in ("box", "centroid", "contours", "prob", "type")} for i in range(lo, hi)]

    def gathers():
        tdist.all_gather_rows(maps, n_total)
        tdist.gather_instance_tables(tables, device)

    return _per_rank(world_size, device, gathers, nbytes=maps.numel() * 4, reps=3,
                     what=("all_gather_rows of the int32 label maps + gather_instance_tables (counts, boxes, centroids, types, "
                           "probabilities, polygon lengths, packed vertices: ragged all-gathers); bytes = label maps only"))


# ----------------------------------------------------------------------------------------------------------------------
def bench_semantic(args) -> dict | None:
    import numpy as np
    import torch

    from tiatoolbox_amd.models.engine.semantic_segmentor import SemanticSegmentor, _finalize, _row_merge
    from tiatoolbox_amd.utils import synth
    from tiatoolbox_amd.wsicore import ArrayWSIReader

    rank, world_size, device = _setup(args)
    side = int(args.slide)
    # synthetic slide: a G-he tile mosaic on a bright background (20 % margin + gutters, so the Otsu mask drops tiles)
    tile = torch.from_numpy(synth.g_he(1, 2048, 2048, seed=3)[0]).to(device)
    slide = torch.full((side, side, 3), 243, dtype=torch.uint8, device=device)
    lo, hi = side // 10, side - side // 10
    for y in range(lo, hi, 2048 + 256):
        for x in range(lo, hi, 2048 + 256):
            h, w = min(2048, hi - y), min(2048, hi - x)
            slide[y:y + h, x:x + w] = tile[:h, :w]
    reader = ArrayWSIReader(slide, mpp=0.25, power=40.0)
    eng = SemanticSegmentor("fcn_resnet50_unet-bcss", batch_size=int(os.environ.get("TIA_SEM_BATCH", "16")),
                            device=str(device), verbose=False)
    result = {}
    import shutil
    import tempfile

    # WSI mode writes its result like the reference does (one file per slide under `save_dir`); /dev/shm keeps the disk out of
    # the measurement, the np.savez of the 400 MB uint8 map stays in
    scratch = Path(tempfile.mkdtemp(prefix="tia_sem_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))

    def step():
        result["out"] = eng.run([reader], patch_mode=False, save_dir=scratch / "out", overwrite=True)

    step()  # lazy loads, weight packing
    cfg = eng._ioconfig  # noqa: SLF001
    mask_reader = reader.tissue_mask(resolution=1.25, units="power")
    in_b, out_b, keep = eng.get_coordinates(reader, mask_reader)
    n_patches, n_grid = int(keep.sum()), len(keep)
    elapsed = _timed(step, args, world_size, device)
    if rank == 0:
        with np.load(result["out"][0]) as saved:
            pred = saved["predictions"]
        assert pred.shape == (side, side) and pred.dtype == np.uint8
    shutil.rmtree(scratch, ignore_errors=True)
    per_rank = semantic_per_rank(out_b, int(cfg.patch_output_shape[0]), side, rank, world_size, device)
    if rank != 0:
        return None
    total = n_patches * args.steps
    ph = int(cfg.patch_input_shape[0])
    oh = int(cfg.patch_output_shape[0])
    # dominant hand-written kernels: one patch row through the canvas kernels (HIP events)
    per_row = int(np.sum((out_b[:, 1] == out_b[0, 1])))
    blocks = torch.rand((per_row, oh, oh, 5), device=device)
    xs = out_b[out_b[:, 1] == out_b[0, 1]][:, 0]
    t_merge = _ev(lambda: _row_merge(blocks, xs, side))
    row, cnt = _row_merge(blocks, xs, side)
    row2, cnt2 = row.clone(), cnt.clone()  # distinct buffers: the two patch rows of a real stitch
    band = torch.zeros((oh, side), dtype=torch.uint8, device=device)
    t_fin = _ev(lambda: _finalize(row, cnt, 0, row2, cnt2, 450, 450, 450 + 450, None, band, y_base=450))
    merge_bytes = blocks.numel() * 4 + row.numel() * 4 + cnt.numel()
    fin_bytes = 2 * 450 * side * 5 * 4 + 2 * 450 * side + 450 * side
    t_gather = _ev(lambda: reader.read_bounds_batch(in_b[keep][:8]))
    gather_bytes = 2 * 8 * ph * ph * 3
    model = eng._inference_model(torch.float32)  # noqa: SLF001
    x = reader.read_bounds_batch(in_b[keep][:1]).float().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    plain = eng.model.module if hasattr(eng.model, "module") else eng.model
    flops = _flops_of(plain, x)  # counted on the plain torch module (the fused copy launches its own kernels)
    t_fwd = _ev(lambda: eng.model.infer_batch(model, reader.read_bounds_batch(in_b[keep][:8]), device=str(device)), reps=3) / 8
    line = {
        "metric": "patches/s (1024x1024x3 in, 512x512x5 out), SemanticSegmentor(fcn_resnet50_unet-bcss) WSI mode",
        "value": round(total / elapsed, 3), "unit": "patches/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "float32", "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[2]: SemanticSegmentor(fcn_resnet50_unet-bcss, seeded random weights).run("
                                f"[one synthetic {side}x{side} in-memory WSI], patch_mode=False), Otsu tissue mask: "
                                f"{n_patches} of {n_grid} grid patches kept"),
                   "slide": side, "patches_per_slide": n_patches, "megapixels_per_s": round(side * side * args.steps / elapsed / 1e6, 1),
                   "parallelism": f"patch rows sharded over {world_size} rank(s), rank-local bands, one all-gather of the uint8 map"},
        "roofline": {
            "kernel": "row_merge_kernel", "bound": "hbm", "achieved": round(merge_bytes / t_merge / 1e9, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(merge_bytes / t_merge / 1e9 / HBM_PEAK_GBS, 4),
            **_traffic("canvas", ("row_merge_kernel",), "row_merge_kernel"),
            "algorithmic_bytes": merge_bytes, "launch_ms": round(t_merge * 1e3, 4),
            "other_kernels": {
                "finalize_kernel": {"bound": "hbm", "achieved": round(fin_bytes / t_fin / 1e9, 1), "unit": "GB/s",
                                    "frac": round(fin_bytes / t_fin / 1e9 / HBM_PEAK_GBS, 4), "launch_ms": round(t_fin * 1e3, 4),
                                    "algorithmic_bytes": fin_bytes, **_traffic("canvas", ("finalize_kernel",), "finalize_kernel"),
                                    "note": ("`frac` is the algorithmic bytes over the kernel time against the HBM peak; the rows it reads were written "
                                             "by the preceding row_merge launch and are served from L2 / Infinity Cache (counter traffic BELOW the "
                                             "algorithmic bytes): a cache-resident rate, not an HBM rate")},
                "gather_patches_kernel": {"bound": "hbm", "achieved": round(gather_bytes / t_gather / 1e9, 1), "unit": "GB/s",
                                          "frac": round(gather_bytes / t_gather / 1e9 / HBM_PEAK_GBS, 4),
                                          "launch_ms": round(t_gather * 1e3, 4), "algorithmic_bytes": gather_bytes,
                                          **_traffic("canvas", ("gather_patches_kernel",), "gather_patches_kernel")}},
            "backbone": {"bound": "mfma", "what": ("UNet-R50 forward per 1024^2 patch, batch 8: " + type(model).__name__
                                                   + (" (every convolution hand-written: stem kernel, 61 on the MFMA kernel with BN / ReLU / "
                                                      "residual fused, class head kernel)" if type(model).__name__ == "FusedUNet" else " (MIOpen)")),
                         "gflop_per_patch": round(flops / 1e9, 1), "achieved": round(flops / t_fwd / 1e12, 2),
                         "peak": MFMA_PEAK_F32, "unit": "TFLOP/s", "frac": round(flops / t_fwd / 1e12 / MFMA_PEAK_F32, 4),
                         "ms_per_patch": round(t_fwd * 1e3, 3)}},
    }
    if per_rank is not None:
        line["per_rank"] = per_rank
        if _RCCL is not None:
            line["rccl"] = _RCCL
    if world_size == 1:
        # the audit mode (conv_algo="direct": no Winograd for the plain 3x3 / stride-1 convolutions of the UNet): forward alone + one slide
        xb8 = reader.read_bounds_batch(in_b[keep][:8])
        ref_logits = eng.model.infer_batch(model, xb8[:2], device=str(device)).clone()
        eng.conv_algo = "direct"
        model_w = eng._inference_model(torch.float32)  # noqa: SLF001
        got_logits = eng.model.infer_batch(model_w, xb8[:2], device=str(device))
        t_fwd_w = _ev(lambda: eng.model.infer_batch(model_w, xb8, device=str(device)), reps=3) / 8
        scratch_w = Path(tempfile.mkdtemp(prefix="tia_semw_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
        t0 = time.perf_counter()
        eng.run([reader], patch_mode=False, save_dir=scratch_w / "out", overwrite=True, conv_algo="direct")
        torch.cuda.synchronize()
        t_w = time.perf_counter() - t0
        shutil.rmtree(scratch_w, ignore_errors=True)
        eng.conv_algo = "auto"
        line["extras"] = {"cnn_direct": {
            "value": round(n_patches / t_w, 3), "unit": "patches/s", "ms_per_step": round(t_w * 1e3, 2),
            "forward_ms_per_patch": round(t_fwd_w * 1e3, 3), "forward_tflops": round(flops / t_fwd_w / 1e12, 2),
            "max_abs_dlogit_vs_default": float((got_logits.float() - ref_logits.float()).abs().max().item()),
            "note": ("conv_algo='direct' (audit mode); `value` runs the default conv_algo='auto': Bottleneck conv2 and the decoder's 3x3 "
                     "convolutions through conv3x3_wino_kernel; one timed slide")}}
    if not args.no_cpu_baseline and world_size == 1:
        from oracle import semantic as osem
        from tiatoolbox_amd.models.architecture import get_pretrained_model

        cpu_model, _ = get_pretrained_model("fcn_resnet50_unet-bcss")
        cpu_model.eval()
        xc = torch.rand(1, 3, ph, ph) * 255
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        with torch.inference_mode():
            cpu_model(xc)
            t_cnn = _median_time(lambda: cpu_model(xc))
        sample = np.random.default_rng(0).random((4, oh, oh, 5), dtype=np.float32)
        locs = np.array([[0, 0, oh, oh], [450, 0, 450 + oh, oh], [0, 450, oh, 450 + oh], [450, 450, 450 + oh, 450 + oh]])
        t_merge_cpu = _median_time(lambda: osem.merge_wsi(sample, locs, (450 + oh, 450 + oh))) / 4
        line["cpu_baseline"] = {
            "value": round(1.0 / (t_cnn + t_merge_cpu), 4), "unit": "patches/s", "cores": min(os.cpu_count() or 1, 64),
            "kind": "port", "repeats": 3, "statistic": "median",
            "sample": (f"1 patch: torch-CPU fp32 UNet-R50 forward ({t_cnn:.2f} s) + oracle overlap-average merge "
                       f"({t_merge_cpu * 1e3:.1f} ms per patch over a 2x2 patch neighbourhood)")}
    return line


# ----------------------------------------------------------------------------------------------------------------------
def bench_hovernet(args) -> dict | None:
    import warnings

    import numpy as np
    import torch

    from tiatoolbox_amd.models.architecture import _hover_device as hd
    from tiatoolbox_amd.models.engine.multi_task_segmentor import NucleusInstanceSegmentor
    from tiatoolbox_amd.utils import synth

    rank, world_size, device = _setup(args)
    n = args.patches if args.patches != 4096 else 256  # tiles per GPU per step
    host = synth.g_he(min(n, 64), 256, 256, seed=5)
    tiles = np.ascontiguousarray(np.tile(host, ((n * world_size + len(host) - 1) // len(host), 1, 1, 1))[:n * world_size])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        eng = NucleusInstanceSegmentor("hovernet_fast-pannuke", batch_size=int(os.environ.get("TIA_HOVER_BATCH", "32")),
                                       device=str(device), verbose=False)
    result = {}

    def step():
        result["out"] = eng.run(tiles, patch_mode=True)

    step()
    elapsed = _timed(step, args, world_size, device)
    out = result["out"]
    assert out["predictions"].shape == (n * world_size, 164, 164)
    per_rank = hovernet_per_rank(out, rank, world_size, device)
    if rank != 0:
        return None
    # post-processing alone, on synthetic head outputs with ~60 nuclei per 164^2 tile (HIP events)
    npm, hv, tp = synth.hover_head_maps(8, 164, 164, seed=1, n_blobs=60)
    reps = max(1, n // 8)
    npm_d = torch.from_numpy(npm).to(device).repeat(reps, 1, 1, 1)
    hv_d = torch.from_numpy(hv).to(device).repeat(reps, 1, 1, 1)
    tp_d = torch.from_numpy(tp).to(device).repeat(reps, 1, 1, 1)
    m = npm_d.shape[0]
    t_proc = _ev(lambda: hd.proc_np_hv(npm_d, hv_d), reps=5)
    model = eng.model.module if hasattr(eng.model, "module") else eng.model
    t_post = _ev(lambda: model.postproc_batch(npm_d, hv_d, tp_d), reps=3)
    inst, _ = hd.proc_np_hv(npm_d[:8], hv_d[:8])
    n_inst = int(sum(len(np.unique(inst[i].cpu().numpy())) - 1 for i in range(8))) / 8
    alg = m * 164 * 164 * 20
    fmodel = eng._inference_model(torch.float32)  # noqa: SLF001
    x = torch.from_numpy(tiles[:8]).to(device).float().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    flops = _flops_of(model, x) / 8  # counted on the plain torch module (the fused copy launches its own kernels)
    xb = torch.from_numpy(tiles[:32]).to(device)
    t_fwd = _ev(lambda: model.infer_batch(fmodel, xb, device=str(device)), reps=3) / 32
    line = {
        "metric": "tiles/s (256x256x3 in, 164x164 out), NucleusInstanceSegmentor(hovernet_fast-pannuke) incl. HIP post-processing",
        "value": round(n * world_size * args.steps / elapsed, 2), "unit": "tiles/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "float32", "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[3]: NucleusInstanceSegmentor(hovernet_fast-pannuke, seeded random weights)"
                                f".run({n} synthetic 256x256x3 tiles per GPU, patch_mode=True): HoVer-Net fast fp32 + "
                                "Sobel-21/energy/markers/watershed + instance tables and contours on the device"),
                   "tiles_per_gpu": n,
                   "parallelism": f"dp{world_size} (tile-sharded; label maps and ragged instance tables all-gathered)"},
        "roofline": {
            "kernel": "hover _proc_np_hv (6 launches: tile-resident labelling / Sobel f64 + energy / marker pipeline, watershed by relaxation)", "bound": "hbm",
            "achieved": round(alg / t_proc / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(alg / t_proc / 1e9 / HBM_PEAK_GBS, 5),
            **_traffic("hover", ("tia::",), "sobel_energy_tile_kernel"), "algorithmic_bytes": alg,
            "launch_ms": round(t_proc * 1e3, 3),
            "workload": (f"{m} synthetic head maps of 164x164 with ~{n_inst:.0f} nuclei each (utils.synth.hover_head_maps), 20 B/px (SURVEY 8(d)); "
                         "NOT the maps `value` post-processes: the engine run feeds a seeded RANDOM-weight HoVer-Net whose noise-like heads give a few "
                         "huge blobs per tile -- a pathological watershed (~10 x slower than on these maps), so `value` understates a trained model"),
            "postproc_incl_tables_ms": round(t_post * 1e3, 3), "postproc_tiles_per_s": round(m / t_post, 1),
            "backbone": {"bound": "mfma", "what": ("HoVer-Net fast forward per 256^2 tile, batch 32: "
                                                   + type(fmodel).__name__ + (" (every convolution hand-written: 105 on the MFMA "
                                                   "kernel, 36 grouped, 3 class heads; BN / ReLU / residual fused)" if type(fmodel).__name__ ==
                                                   "FusedHoVerNet" else " (MIOpen)")),
                         "gflop_per_tile": round(flops / 1e9, 1), "achieved": round(flops / t_fwd / 1e12, 2),
                         "peak": MFMA_PEAK_F32, "unit": "TFLOP/s", "frac": round(flops / t_fwd / 1e12 / MFMA_PEAK_F32, 4),
                         "ms_per_tile": round(t_fwd * 1e3, 3)}},
    }
    if per_rank is not None:
        line["per_rank"] = per_rank
        if _RCCL is not None:
            line["rccl"] = _RCCL
    if world_size == 1:
        # the audit mode (conv_algo="direct": no Winograd for the plain 3x3 / stride-1 convolutions): an extra beside `value`
        heads_d = [h.clone() for h in model.infer_batch(fmodel, xb, device=str(device))]
        eng.run(tiles, patch_mode=True, conv_algo="direct")
        t0 = time.perf_counter()
        eng.run(tiles, patch_mode=True, conv_algo="direct")
        torch.cuda.synchronize()
        t_w = time.perf_counter() - t0
        fmodel_w = eng._inference_model(torch.float32)  # noqa: SLF001
        heads_w = model.infer_batch(fmodel_w, xb, device=str(device))
        t_fwd_w = _ev(lambda: model.infer_batch(fmodel_w, xb, device=str(device)), reps=3) / 32
        eng.conv_algo = "auto"
        line["extras"] = {"cnn_direct": {
            "value": round(n / t_w, 2), "unit": "tiles/s", "ms_per_step": round(t_w * 1e3, 2),
            "forward_ms_per_tile": round(t_fwd_w * 1e3, 3), "forward_tflops": round(flops / t_fwd_w / 1e12, 2),
            "max_abs_dhead_vs_default": float(max((a.float() - b.float()).abs().max().item() for a, b in zip(heads_w, heads_d))),
            "note": ("conv_algo='direct' (audit mode); `value` runs the default conv_algo='auto': the residual units' 3x3 convolutions "
                     "through conv3x3_wino_kernel; one timed step")}}
    if not args.no_cpu_baseline and world_size == 1:
        from tiatoolbox_amd.models.architecture import get_pretrained_model

        def cpu_post():
            for i in range(4):
                from oracle import hovernet as oh  # (the CPU baseline leg: the only place the oracle runs)

                inst_i = oh.proc_np_hv(npm[i], hv[i])
                oh.get_instance_info(inst_i, np.around(tp[i]).astype("uint8")[..., 0])

        t_post_cpu = _median_time(cpu_post) / 4
        cpu_model, _ = get_pretrained_model("hovernet_fast-pannuke")
        cpu_model.eval()
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        xc = torch.from_numpy(tiles[:4]).float().permute(0, 3, 1, 2)
        with torch.inference_mode():
            cpu_model(xc)
            t_cnn = _median_time(lambda: cpu_model(xc)) / 4
        line["cpu_baseline"] = {
            "value": round(1.0 / (t_cnn + t_post_cpu), 4), "unit": "tiles/s", "cores": min(os.cpu_count() or 1, 64),
            "kind": "port", "repeats": 3, "statistic": "median",
            "sample": (f"4 tiles: torch-CPU fp32 HoVer-Net fast ({t_cnn * 1e3:.0f} ms/tile, threaded) + oracle "
                       f"_proc_np_hv + get_instance_info on one core ({t_post_cpu * 1e3:.0f} ms/tile, ~60 nuclei)")}
    return line


# ----------------------------------------------------------------------------------------------------------------------
def bench_vahadane(args) -> dict | None:
    import numpy as np
    import torch

    from tiatoolbox_amd.tools import _stain_device as dev
    from tiatoolbox_amd.tools.stainaugment import StainAugmentor
    from tiatoolbox_amd.tools.stainnorm import get_normalizer
    from tiatoolbox_amd.utils import synth

    rank, world_size, device = _setup(args)
    n = args.patches if args.patches != 4096 else 8192  # per GPU: 65 536 over 8
    hw = 256
    host = synth.g_he(256, hw, hw, seed=11 + rank)
    x = torch.from_numpy(host).to(device).repeat((n + 255) // 256, 1, 1, 1)[:n].contiguous()
    target = np.load(ROOT / "tests" / "golden" / "target_crop_256.npy")
    norm = get_normalizer("vahadane")
    norm.precision = args.precision
    norm.fit(target)
    aug = StainAugmentor(method="vahadane", sigma1=0.4, sigma2=0.2, augment_background=False,
                         precision="f32" if args.precision == "f32" else "f64")
    rng = np.random.default_rng(0)
    ab = np.concatenate([rng.uniform(0.6, 1.4, (n, 2)), rng.uniform(-0.2, 0.2, (n, 2))], axis=1)
    result = {}

    def step():
        normed = norm.transform(x)                 # per-patch dictionary learning + normalisation, uint8 out
        aug.fit(normed, threshold=0.85)            # per-patch dictionary learning of the normalised patch + statistics
        result["out"] = aug.augment(alpha_beta=ab)

    step()
    elapsed = _timed(step, args, world_size, device)
    out = result["out"]
    assert out.shape == x.shape and out.dtype == torch.uint8
    per_rank = _per_rank(world_size, device, None, what="none (patch-sharded, the normalised patches stay on their rank)")
    if rank != 0:
        return None
    params = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
    t_stats = _ev(lambda: dev.stain_stats(x, params), reps=3)
    px = n * hw * hw
    line = {
        "metric": "patches/s (256x256x3), VahadaneNormalizer.transform + StainAugmentor(vahadane).fit/augment",
        "value": round(n * world_size * args.steps / elapsed, 2), "unit": "patches/s", "n_gpus": world_size,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[4]: VahadaneNormalizer (dictionary learning on the device, sklearn "
                                f"DictionaryLearning restated) + StainAugmentor over {n} synthetic 256x256x3 patches per GPU "
                                f"(65 536 over 8 GPUs); statistics and dictionary f64, per-pixel {args.precision}; BASELINE's 'fp16 OD path' is served "
                                f"as float32 / float64 ARITHMETIC with half as an OUTPUT format only (extras.per_pixel_float32): no fp16 optical densities"),
                   "patches_per_gpu": n, "parallelism": f"dp{world_size} (patch-sharded, no collective)"},
        "roofline": {"kernel": "vahadane_dl_kernel + stain_stats_kernel<false> (Vahadane statistics: dictionary learning by replay, then the common tail)",
                     "bound": "hbm", "achieved": round(px * 3 / t_stats / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(px * 3 / t_stats / 1e9 / HBM_PEAK_GBS, 5),
                     **_traffic("vahadane", ("vahadane_dl_kernel", "stain_stats_kernel"), "vahadane_dl_kernel"),
                     "algorithmic_bytes": px * 3, "launch_ms": round(t_stats * 1e3, 3),
                     "note": ("8 sweeps of float64 arithmetic per pixel with NO dictionary in memory (the atom values are replayed from "
                              "per-iteration scalars): arithmetic-bound, not a bandwidth kernel -- the one-kernel form that keeps the "
                              "2 x N dictionary in HBM (dl_one_kernel) takes 1.8x as long")},
    }
    if per_rank is not None:
        line["per_rank"] = per_rank
        if _RCCL is not None:
            line["rccl"] = _RCCL
    # BASELINE's "fp16 OD path": the per-pixel arithmetic exists in float64 (the reference's, reported above) and float32; half
    # precision exists as an OUTPUT format of the float32 path (the CNN's input), not as OD-space arithmetic -- 11 significand
    # bits cannot hold exp(-OD) to the 1e-4 the north star asks of normalised pixels.  Reported here: the float32 per-pixel path
    # with its measured deviation from the float64 result.
    if world_size == 1 and args.precision != "f32":
        norm32 = get_normalizer("vahadane")
        norm32.precision = "f32"
        norm32.fit(target)
        aug32 = StainAugmentor(method="vahadane", sigma1=0.4, sigma2=0.2, augment_background=False, precision="f32")

        def step32():
            normed = norm32.transform(x)
            aug32.fit(normed, threshold=0.85)
            result["out32"] = aug32.augment(alpha_beta=ab)

        step32()
        t32 = _ev(step32, reps=max(1, min(3, args.steps)))
        ref64 = norm.transform(x[:512])
        got32 = norm32.transform(x[:512])
        diff = (ref64.to(torch.int16) - got32.to(torch.int16)).abs()
        unit16 = norm32.transform(x[:512], out="unit_float16").float()
        line["extras"] = {"per_pixel_float32": {
            "value": round(n / t32, 2), "unit": "patches/s", "ms_per_step": round(t32 * 1e3, 2),
            "max_abs_byte_diff_vs_float64": int(diff.max()), "differing_bytes_fraction": round(float((diff != 0).float().mean()), 8),
            "max_abs_unit_float16_vs_float64_bytes_over_255": round(float((unit16.reshape(ref64.shape) - ref64.float() / 255.0).abs().max()), 6),
            "note": ("extra only: normalise + augment with float32 per-pixel arithmetic (statistics and dictionary stay float64); "
                     "'fp16 OD path' of BASELINE configs[4] = this path with out='unit_float16' (half as output format only)")}}
    if not args.no_cpu_baseline and world_size == 1:
        from oracle import stain as ostain

        ref = ostain.get_normalizer("vahadane")
        ref.fit(target.copy())

        vex = ostain.VahadaneExtractor(random_state=0)

        def cpu_step():
            for p in host[:2]:
                nrm = ref.transform(p.copy())
                sm = vex.get_stain_matrix(nrm.copy())  # StainAugmentor.fit: stain matrix of the image to augment
                ostain.stain_augment(nrm, sm, np.array([1.1, 0.9]), np.array([0.05, -0.05]), threshold=0.85)

        cpu_step()
        t_cpu = _median_time(cpu_step) / 2
        # all cores: one single-threaded worker per core over a bounded sample (BASELINE.md asks for both figures)
        import multiprocessing as mp
        import time

        cores = min(os.cpu_count() or 1, 256)
        thread_vars = ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS")
        saved = {k: os.environ.get(k) for k in thread_vars}
        os.environ.update(dict.fromkeys(thread_vars, "1"))
        sample = [host[i % len(host)] for i in range(2 * cores)]
        try:
            with mp.get_context("spawn").Pool(cores) as pool:
                pool.map(_vahadane_cpu_worker, [(target, host[0])] * cores)  # start the workers, fit the target once each
                t0 = time.perf_counter()
                pool.map(_vahadane_cpu_worker, [(target, q) for q in sample], chunksize=1)
                t_all = time.perf_counter() - t0
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        line["cpu_baseline"] = {"value": round(len(sample) / t_all, 3), "unit": "patches/s", "cores": cores, "kind": "port",
                                "sample": (f"{len(sample)} patches over a {cores}-process pool (one thread each): oracle Vahadane "
                                           "transform (scikit-learn DictionaryLearning, as the reference) + oracle StainAugmentor "
                                           "fit / augment"),
                                "one_core": {"value": round(1.0 / t_cpu, 4), "unit": "patches/s", "cores": 1, "repeats": 3,
                                             "statistic": "median", "sample": "2 patches on one core, same two stages"}}
    return line
