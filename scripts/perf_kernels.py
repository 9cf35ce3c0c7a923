"""Stage-level roofline measurements of the mask / Reinhard / HoVer-Net post-processing / canvas kernels.

HIP-event timing (torch's current stream = the stream the C ABI launches on), 3 warm-ups + 20 reps.
``achieved`` = SURVEY section 8(d) algorithmic bytes / stage time; peak = 8 TB/s.  Prints one JSON line per
stage; run under ``rocprofv3 --kernel-trace --stats`` for the per-kernel split.
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.models.architecture import _hover_device as hd
from tiatoolbox_amd.models.engine import semantic_segmentor as ss
from tiatoolbox_amd.tools import _img_device as img
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.tools import reinhard as rh
from tiatoolbox_amd.tools.tissuemask import MorphologicalMasker, OtsuTissueMasker
from tiatoolbox_amd.utils import synth

PEAK = 8000.0
only = set(sys.argv[1:])


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(stage, ms, alg_bytes, **extra):
    gbs = alg_bytes / ms / 1e6
    print(json.dumps({"stage": stage, "ms": round(ms, 4), "alg_MB": round(alg_bytes / 1e6, 2), "GBps": round(gbs, 1),
                      "frac_hbm": round(gbs / PEAK, 4), **extra}), flush=True)


def want(name):
    return not only or name in only


def patches(n, h, w):
    base = torch.from_numpy(synth.g_he(64, h, w, seed=1)).cuda()
    return base.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()


def sec_reinhard():
    n, h, w = 4096, 224, 224
    x = patches(n, h, w)
    norm = rh.ReinhardNormalizer()
    norm.fit(x[0])
    out = norm.transform(x)
    report("reinhard.transform (lab_hist + fused apply), 4096x224^2", timeit(lambda: norm.transform(x)), 2 * x.numel())
    report("lab_hist only", timeit(lambda: norm._lab_hist(x)), x.numel())  # noqa: SLF001
    report("rgb->lab convert only", timeit(lambda: rh.lab_convert(x, 0)), 2 * x.numel())
    del x, out


def sec_mask():
    n, h, w = 16, 2048, 2048   # slide thumbnails (20k^2 WSI at 1.25x is ~1250^2; a batch of larger ones)
    x = torch.from_numpy(synth.g_he(n, 512, 512, seed=2)).cuda().repeat_interleave(4, 1).repeat_interleave(4, 2).contiguous()
    px = n * h * w
    om = OtsuTissueMasker()
    om.fit(x)
    report("otsu.fit (rgb2gray + hist256), 16x2048^2", timeit(lambda: om.fit(x)), 3 * px)
    report("otsu.transform (fused gray+threshold)", timeit(lambda: om.transform(x)), 4 * px)
    mm = MorphologicalMasker(power=1.25)
    mm.fit(x)
    report(f"morphological.transform (threshold + CCL8 + area filter + dilate {mm.kernel_size})",
           timeit(lambda: mm.transform(x), reps=5, warm=1), 4 * px)
    m = om._masks(x)  # noqa: SLF001
    report("  ccl_label(8)", timeit(lambda: img.ccl_label(m, connectivity=8), reps=5, warm=1), 5 * px)
    lab, _ = img.ccl_label(m, connectivity=8)
    report("  binary_morph dilate", timeit(lambda: img.binary_morph(m, img.offsets_of(mm.kernel, m.device), "dilate"), reps=5, warm=1), 2 * px)
    report("  fill_holes", timeit(lambda: img.fill_holes(m), reps=5, warm=1), 2 * px)
    del x, m, lab


def sec_hover():

    n, h, w = 256, 164, 164
    npm, hv, tp = synth.hover_head_maps(8, h, w, seed=4, n_blobs=60)
    rep = n // 8
    npm_t = torch.from_numpy(npm).cuda().repeat(rep, 1, 1, 1)
    hv_t = torch.from_numpy(hv).cuda().repeat(rep, 1, 1, 1)
    tp_t = torch.from_numpy(np.around(tp).astype("uint8")[..., 0]).cuda().repeat(rep, 1, 1)
    px = n * h * w
    inst, nmark = hd.proc_np_hv(npm_t, hv_t)
    mx = int(nmark.max())
    report("hover proc_np_hv (Sobel21 .. watershed), 256x164^2", timeit(lambda: hd.proc_np_hv(npm_t, hv_t), reps=5, warm=1),
           20 * px, instances=int(nmark.sum()))
    report("hover instance_stats", timeit(lambda: hd.instance_stats(inst, tp_t, mx, 6), reps=5, warm=1), 5 * px)
    stats, _ = hd.instance_stats(inst, tp_t, mx, 6)
    report("hover contours (scan + write, incl. D2H)", timeit(lambda: hd.contours(inst, stats, mx), reps=5, warm=1), 4 * px)
    # WSI-mode tile
    npm1, hv1, _ = synth.hover_head_maps(1, 1000, 1000, seed=5, n_blobs=1500)
    a, b = torch.from_numpy(npm1).cuda(), torch.from_numpy(hv1).cuda()
    i1, n1 = hd.proc_np_hv(a, b)
    report("hover proc_np_hv, one 1000^2 tile", timeit(lambda: hd.proc_np_hv(a, b), reps=5, warm=1), 20 * 1000 * 1000,
           instances=int(n1.sum()))
    # head maps that look like noise (what a random-weight network emits): a few huge blobs per tile
    from scipy import ndimage

    rng = np.random.default_rng(0)
    npn = np.stack([ndimage.gaussian_filter(rng.standard_normal((h, w)), 5.0) for _ in range(8)])
    npn = (npn / np.abs(npn).max() * 0.5 + 0.55).astype(np.float32)[..., None]
    hvn = np.stack([ndimage.gaussian_filter(rng.standard_normal((h, w, 2)), (3.0, 3.0, 0)) for _ in range(8)])
    hvn = (hvn / np.abs(hvn).max()).astype(np.float32)
    an, bn = torch.from_numpy(npn).cuda().repeat(rep, 1, 1, 1), torch.from_numpy(hvn).cuda().repeat(rep, 1, 1, 1)
    i2, n2 = hd.proc_np_hv(an, bn)
    report("hover proc_np_hv, 256x164^2 noise-like maps (huge blobs)", timeit(lambda: hd.proc_np_hv(an, bn), reps=3, warm=1),
           20 * px, instances=int(n2.sum()))


def sec_canvas():
    n, oh_, ow, c = 40, 512, 512, 5
    stride = 450
    width = stride * (n - 1) + ow
    blocks = torch.rand((n, oh_, ow, c), device="cuda")
    xs = np.arange(n) * stride
    row, cnt = ss._row_merge(blocks, xs, width)  # noqa: SLF001
    report("canvas row_merge, 40 x 512^2 x 5 f32", timeit(lambda: ss._row_merge(blocks, xs, width)),  # noqa: SLF001
           blocks.numel() * 4 + row.numel() * 4 + cnt.numel())
    probs = torch.empty((oh_, width, c), device="cuda")
    pred = torch.empty((oh_, width), dtype=torch.uint8, device="cuda")
    report("canvas finalize (A+B, /count, argmax)",
           timeit(lambda: ss._finalize(row, cnt, 0, row, cnt, 450, 0, 450, probs, pred)),  # noqa: SLF001
           450 * width * (c * 4 * 2 + 1 + 1))


def sec_stain_misc():
    n, h, w = 4096, 224, 224
    x = patches(n, h, w)
    from tiatoolbox_amd.tools.stainaugment import StainAugmentor
    from tiatoolbox_amd.tools.stainnorm import get_normalizer

    norm = get_normalizer("macenko")
    norm.fit(x[0])
    p = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
    stats = dev.stain_stats(x, p)
    ab = torch.rand((n, 4), device="cuda", dtype=torch.float64) * 0.2 + 0.9
    report("stain_augment, 4096x224^2", timeit(lambda: dev.augment(x, stats, ab, p.y_thr, augment_background=False, zero_to_one=False)),
           2 * x.numel())
    report("stain_augment f32 / 16-byte accesses", timeit(lambda: dev.augment(x, stats, ab, p.y_thr, augment_background=False,
                                                                        zero_to_one=False, math=_lib.MATH_F32)), 2 * x.numel())
    report("luminosity_mask", timeit(lambda: dev.luminosity_mask(x, stats, p.y_thr)), x.numel() * 4 // 3)


import traceback

for _name in ('reinhard', 'mask', 'hover', 'canvas', 'stain_misc'):
    if want(_name):
        try:
            globals()["sec_" + _name]()
        except Exception:  # noqa: BLE001  (keep measuring the other stages)
            traceback.print_exc()
        torch.cuda.empty_cache()
