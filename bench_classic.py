#!/usr/bin/env python
"""Stage-level rooflines of the HBM-bound "classic" kernels of the hot path: Reinhard (Lab statistics + transform), the Otsu /
morphological tissue maskers, the luminosity mask, stain augmentation, the 8-bit Lab conversion.

``classic(...)`` returns one entry per stage -- ``bench.py`` stores it as ``extras.classic`` so that the driver's own bench run
observes these numbers -- and the module runs on its own for kernel work:

    python bench_classic.py [stage ...] [--reps N] [--calls N]      # stages: reinhard mask luminosity augment
    python bench_classic.py --pmc reinhard                          # workload of a rocprofv3 --pmc pass (a few calls, no timing)

Per stage: HIP-event time on the launch stream (the C ABI launches on torch's current stream; 3 warm-ups, ``reps`` timed calls,
mean), ``algorithmic_bytes`` as section 8(d) defines them, ``achieved`` = bytes / time against the 8 TB/s HBM peak, ``traffic`` = HBM-side
bytes per call from this round's committed ``rocprofv3 --pmc FETCH_SIZE`` / ``WRITE_SIZE`` passes over the same workload
(``profiles/*_classic_<stage>_pmc_*.txt``; null when no pass is committed), and a one-line CPU baseline: the oracle (``oracle/``, the
NumPy restatement of the reference) on a bounded sample of the same input, one process (``cores`` says so).
"""

from __future__ import annotations

import json
import os
import re
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0
STAGES = ("reinhard", "mask", "luminosity", "augment")


def _ev_time(fn, reps: int = 20, warm: int = 3) -> float:
    """Mean seconds per call, HIP events on the current (= launch) stream."""
    import torch

    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def _traffic(stem: str, kernels: tuple[str, ...], per_call_kernel: str) -> dict | None:
    """HBM-side bytes per call from the committed counter passes (gfx950: FETCH_SIZE counts half the bytes read; units KiB)."""
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted((ROOT / "profiles").glob(f"*_{stem}_pmc_{counter}.txt"))
        if not files:
            return None
        tot, calls = 0.0, 0
        for line in files[-1].read_text().splitlines():
            m = re.search(rf"{counter} mean=\s*([0-9.]+) n=\s*(\d+)\s+(.*)", line)
            if not m:
                continue
            if any(k in m.group(3) for k in kernels):
                tot += float(m.group(1)) * int(m.group(2))
            if per_call_kernel in m.group(3):
                calls += int(m.group(2))
        if calls == 0:
            return None
        vals[counter] = tot / calls * 1024.0
        vals["file_" + counter] = files[-1].name
    return {"bytes": round(2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]),
            "source": f"profiles/{vals['file_FETCH_SIZE']} (x2, gfx950) + profiles/{vals['file_WRITE_SIZE']}; totals over {list(kernels)} "
                      f"per dispatch of {per_call_kernel}"}


def _entry(what: str, seconds: float, alg_bytes: int, *, kernels: tuple[str, ...] = (), stem: str | None = None,
           per_call: str | None = None, cpu: dict | None = None, **extra) -> dict:
    gbs = alg_bytes / seconds / 1e9
    out = {"what": what, "bound": "hbm", "launch_ms": round(seconds * 1e3, 4), "algorithmic_bytes": int(alg_bytes),
           "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None}
    if kernels:
        out["kernels"] = list(kernels)
    if stem and kernels:
        t = _traffic(stem, kernels, per_call or kernels[0])
        if t:
            out["traffic"] = t["bytes"]
            out["traffic_source"] = t["source"]
    if cpu:
        out["cpu_baseline"] = cpu
    out.update(extra)
    return out


def _cpu(fn, units: int, unit: str, sample: str, reps: int = 3) -> dict:
    """Median wall time of the oracle call ``fn`` (one process): ``units`` per call -> units/s."""
    import statistics

    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return {"value": round(units / statistics.median(ts), 3), "unit": unit, "cores": 1, "kind": "port", "sample": sample}


def _patches(n: int, h: int, w: int, seed: int = 1):
    import torch

    from tiatoolbox_amd.utils import synth

    host = synth.g_he(64, h, w, seed=seed)
    x = torch.from_numpy(host).cuda().repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
    return host, x


def _thumbnails(n: int, side: int):
    """Slide-thumbnail-like RGB images for the maskers: G-he patches of side/4 upsampled x4 (tissue blobs a few dozen pixels wide)."""
    import torch

    from tiatoolbox_amd.utils import synth

    host = synth.g_he(n, side // 4, side // 4, seed=2).repeat(4, axis=1).repeat(4, axis=2)
    return host, torch.from_numpy(host).cuda().contiguous()


# ----------------------------------------------------------------------------------------------------------------------
def stage_reinhard(reps: int, cpu: bool, pmc_calls: int = 0) -> dict:
    import numpy as np

    from tiatoolbox_amd.tools import reinhard as rh

    n, h, w = 4096, 224, 224
    host, x = _patches(n, h, w)
    target = np.load(ROOT / "tests" / "golden" / "target_crop_256.npy")
    norm = rh.ReinhardNormalizer()
    norm.fit(target)
    if pmc_calls:
        for _ in range(pmc_calls):
            norm.transform(x)
            norm.lab_statistics(x)
            rh.lab_convert(x, 0)
        return {}
    out = {}
    base = None
    if cpu:
        from oracle import stain as ostain

        ref = ostain.get_normalizer("reinhard")
        ref.fit(target.copy())
        sub = [host[i].copy() for i in range(8)]
        base = _cpu(lambda: [ref.transform(p) for p in sub], len(sub), "patches/s",
                    "oracle ReinhardNormalizer.transform on 8 of the 224x224 patches")
    out["reinhard_transform"] = _entry(
        f"ReinhardNormalizer.transform on {n} x {h}x{w}x3 uint8 patches resident in HBM: Lab statistics + per-patch tables + "
        "RGB->Lab->table->RGB (stainnorm.py:222-367); bytes = read u8 + write u8",
        _ev_time(lambda: norm.transform(x), reps), 2 * x.numel(), kernels=("reinhard_resident_kernel<13, false>",), stem="classic_reinhard",
        cpu=base, note="one launch; bound by the vector ALU and LDS look-ups (~63 VALU instructions and ~70 LDS cycles per 64 pixels), not by HBM")
    out["lab_statistics"] = _entry(
        f"Lab mean / std of {n} x {h}x{w} patches (get_mean_std, stainnorm.py:263-279: RGB->Lab + per-channel moments); bytes = read u8",
        _ev_time(lambda: norm.lab_statistics(x), reps), x.numel(), kernels=("reinhard_resident_kernel<13, true>",), stem="classic_reinhard")
    out["lab_convert"] = _entry(
        f"cv2.cvtColor(RGB2LAB), 8-bit, {n} x {h}x{w}; bytes = read u8 + write u8",
        _ev_time(lambda: rh.lab_convert(x, 0), reps), 2 * x.numel(), kernels=("lab_stream_kernel<1>",), stem="classic_reinhard")
    return out


def stage_mask(reps: int, cpu: bool, pmc_calls: int = 0) -> dict:
    from tiatoolbox_amd.tools.tissuemask import MorphologicalMasker, OtsuTissueMasker

    n, side = 16, 2048
    host, x = _thumbnails(n, side)
    px = n * side * side
    om = OtsuTissueMasker()
    om.fit(x)
    mm = MorphologicalMasker(power=1.25)
    mm.fit(x)
    if pmc_calls:
        for _ in range(pmc_calls):
            om.fit(x)
            om.transform(x)
            mm.transform(x)
        return {}
    out = {}
    b_fit = b_tr = b_mm = None
    if cpu:
        from oracle import tissuemask as otm

        ro, rm = otm.OtsuTissueMasker(), otm.MorphologicalMasker(power=1.25)
        one = host[:1]
        ro.fit(one)
        rm.fit(one)
        b_fit = _cpu(lambda: ro.fit(one), side * side / 1e6, "Mpx/s", f"oracle OtsuTissueMasker.fit on one {side}^2 image")
        b_tr = _cpu(lambda: ro.transform(one), side * side / 1e6, "Mpx/s", f"oracle OtsuTissueMasker.transform on one {side}^2 image")
        b_mm = _cpu(lambda: rm.transform(one), side * side / 1e6, "Mpx/s", f"oracle MorphologicalMasker.transform on one {side}^2 image", reps=1)
    out["otsu_fit"] = _entry(
        f"OtsuTissueMasker.fit on {n} x {side}^2 RGB thumbnails (grey + 256-bin histogram + threshold, tissuemask.py:99-137); bytes = read u8 RGB",
        _ev_time(lambda: om.fit(x), reps), 3 * px, kernels=("gray_hist_kernel", "otsu_threshold_kernel"), stem="classic_mask",
        per_call="gray_hist_kernel", cpu=b_fit, note="grey + histogram in one pass, Otsu's arithmetic on the 256 counts on the device; the "
        "threshold stays on the device until the attribute is read")
    out["otsu_transform"] = _entry(
        f"OtsuTissueMasker.transform, same images (grey < threshold, tissuemask.py:139-164); bytes = read RGB + write mask",
        _ev_time(lambda: om.transform(x), reps), 4 * px, kernels=("threshold_wide_kernel",), stem="classic_mask", cpu=b_tr)
    out["morphological_transform"] = _entry(
        f"MorphologicalMasker(power=1.25).transform, same images (threshold + 8-connected small-region removal below "
        f"{mm.min_region_size} px + {tuple(int(k) for k in mm.kernel_size)} elliptical dilation, tissuemask.py:270-306); bytes = read RGB + write mask",
        _ev_time(lambda: mm.transform(x), max(3, reps // 4), 1), 4 * px,
        kernels=("morph_mask_tile_kernel",), stem="classic_mask", cpu=b_mm,
        note="one launch: 256 x 128 tiles labelled in LDS (bit-row flood from >= min-size certificates, union-find for the rest)")
    return out


def stage_luminosity(reps: int, cpu: bool, pmc_calls: int = 0) -> dict:
    import numpy as np

    from tiatoolbox_amd.tools import _stain_device as dev
    from tiatoolbox_amd.tools.stainnorm import get_normalizer

    n, h, w = 4096, 224, 224
    host, x = _patches(n, h, w)
    norm = get_normalizer("macenko")
    norm.fit(np.load(ROOT / "tests" / "golden" / "target_crop_256.npy"))
    p = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
    stats = dev.stain_stats(x, p)
    if pmc_calls:
        for _ in range(pmc_calls):
            dev.luminosity_mask(x, stats, p.y_thr)
        return {}
    base = None
    if cpu:
        from oracle import stain as ostain

        sub = [host[i].copy() for i in range(8)]
        base = _cpu(lambda: [ostain.get_luminosity_tissue_mask(s, 0.8) for s in sub], len(sub), "patches/s",
                    "oracle get_luminosity_tissue_mask (contrast_enhancer + 8-bit Lab + threshold) on 8 of the patches")
    return {"luminosity_mask": _entry(
        f"get_luminosity_tissue_mask of {n} x {h}x{w} patches given their percentiles (contrast stretch + Lab L < 0.8, misc.py:261-290); "
        "bytes = read RGB + write mask", _ev_time(lambda: dev.luminosity_mask(x, stats, p.y_thr), reps), x.numel() * 4 // 3,
        kernels=("luminosity_mask_wide_kernel",), stem="classic_luminosity", cpu=base)}


def stage_augment(reps: int, cpu: bool, pmc_calls: int = 0) -> dict:
    import numpy as np
    import torch

    from tiatoolbox_amd import _lib
    from tiatoolbox_amd.tools import _stain_device as dev
    from tiatoolbox_amd.tools.stainnorm import get_normalizer

    n, h, w = 4096, 224, 224
    host, x = _patches(n, h, w)
    norm = get_normalizer("macenko")
    norm.fit(np.load(ROOT / "tests" / "golden" / "target_crop_256.npy"))
    p = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
    stats = dev.stain_stats(x, p)
    ab = torch.rand((n, 4), device="cuda", dtype=torch.float64) * 0.2 + 0.9

    def aug(math):
        return dev.augment(x, stats, ab, p.y_thr, augment_background=False, zero_to_one=False, math=math)

    if pmc_calls:
        for _ in range(pmc_calls):
            aug(_lib.MATH_F64)
            aug(_lib.MATH_F32)
        return {}
    base = None
    if cpu:
        from oracle import stain as ostain

        sm = np.asarray(norm.stain_matrix_target, dtype=np.float64)
        sub = [host[i].copy() for i in range(4)]
        base = _cpu(lambda: [ostain.stain_augment(s, sm, np.array([1.05, 0.95]), np.array([0.01, -0.01])) for s in sub], len(sub), "patches/s",
                    "oracle stain_augment (concentrations by lstsq + recomposition, given stain matrix) on 4 of the patches")
    return {
        "augment_f64": _entry(
            f"StainAugmentor.apply on {n} x {h}x{w} patches given their statistics, the reference's float64 per-pixel arithmetic "
            "(stainaugment.py:177-206); bytes = read u8 + write u8", _ev_time(lambda: aug(_lib.MATH_F64), reps), 2 * x.numel(),
            kernels=("stain_augment_f64_wide_kernel",), stem="classic_augment", cpu=base,
            note="product of per-patch float64 tables (the augmented optical density is affine in the input optical densities)"),
        "augment_f32": _entry(
            "same call, precision='f32' (opt-in: float32 per-pixel arithmetic with the hardware exp2, 16-byte accesses)",
            _ev_time(lambda: aug(_lib.MATH_F32), reps), 2 * x.numel(), kernels=("stain_augment_wide_kernel",), stem="classic_augment"),
    }


def classic(stages: tuple[str, ...] = STAGES, reps: int = 20, cpu: bool = True) -> dict:
    """Every stage's entries in one dict; a stage that fails reports ``{"error": ...}`` instead of taking the run down."""
    import gc

    import torch

    out: dict = {}
    for name in stages:
        try:
            out.update(globals()[f"stage_{name}"](reps, cpu))
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        gc.collect()
        torch.cuda.empty_cache()
    return out


def main() -> None:
    args = sys.argv[1:]
    reps, calls, pmc = 20, 3, False
    stages = []
    i = 0
    while i < len(args):
        if args[i] == "--reps":
            reps = int(args[i + 1])
            i += 2
        elif args[i] == "--calls":
            calls = int(args[i + 1])
            i += 2
        elif args[i] == "--pmc":
            pmc = True
            i += 1
        elif args[i] == "--no-cpu":
            os.environ["TIA_CLASSIC_NO_CPU"] = "1"
            i += 1
        else:
            stages.append(args[i])
            i += 1
    stages = tuple(stages) or STAGES
    if pmc:
        import torch

        for s in stages:
            globals()[f"stage_{s}"](0, False, pmc_calls=calls)
        torch.cuda.synchronize()
        print(f"PMC calls={calls} stages={','.join(stages)}")
        return
    res = classic(stages, reps, cpu=not os.environ.get("TIA_CLASSIC_NO_CPU"))
    for k, v in res.items():
        short = {kk: v[kk] for kk in ("launch_ms", "achieved", "frac", "traffic", "error") if kk in v}
        print(f"{k:28s} {json.dumps(short)}", file=sys.stderr)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
