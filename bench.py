#!/usr/bin/env python
"""Headline benchmark: Macenko stain normalisation + resnet18 PatchPredictor, patches/s.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: 4096 synthetic 224x224x3 uint8 patches per GPU (G-he,
seeded), Macenko fitted on a target crop, resnet18-kather100k architecture with seeded random
weights (pretrained weights are unreachable offline).  One *step* = one pass of the hot path
over the GPU's 4096 resident patches: per-patch Macenko statistics (HIP), fused
normalise->ToTensor apply (HIP), resnet18 forward (MIOpen/hipBLASLt through PyTorch-ROCm),
softmax, argmax, and (N>1) the RCCL all-gather of the per-patch probabilities.  Inputs are
already resident in HBM when the timed region starts.  Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = {"float16": 2500.0, "bfloat16": 2500.0, "float32": 157.3}
RESNET18_GFLOP_224 = 3.64  # 1.82 GMAC per 224x224 patch


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patches", type=int, default=4096, help="patches per GPU per step")
    ap.add_argument("--patch-size", type=int, default=224)
    ap.add_argument("--micro-batch", type=int, default=1024, help="CNN forward batch")
    ap.add_argument("--dtype", default=os.environ.get("TIA_BENCH_DTYPE", "float16"),
                    choices=["float32", "float16", "bfloat16"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"],
                    help="per-pixel arithmetic of the stain apply kernel (statistics are always f64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64)
    return ap.parse_args()


def cpu_norm_worker(args):
    """Oracle Macenko transform of one patch (runs in a worker process)."""
    import numpy as np

    from oracle import stain as ostain

    target, patch = args
    norm = cpu_norm_worker.cache.get("n")
    if norm is None:
        norm = ostain.get_normalizer("macenko")
        norm.fit(np.array(target))
        cpu_norm_worker.cache["n"] = norm
    return norm.transform(np.array(patch))


cpu_norm_worker.cache = {}


def cpu_baseline(target, patches, model_cpu, sample: int) -> dict:
    """The CPU oracle (NumPy restatement of the reference path) + torch-CPU fp32 resnet18 on a
    bounded sample of the same workload, all host cores."""
    import multiprocessing as mp

    import numpy as np
    import torch

    cores = os.cpu_count() or 1
    sample = min(max(sample, 4 * min(cores, 256)), len(patches))
    cnn_threads = min(cores, 64)
    sub = [np.ascontiguousarray(p) for p in patches[:sample]]
    # one BLAS/OpenMP thread per worker process: the pool already uses every core
    thread_vars = ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in thread_vars}
    os.environ.update(dict.fromkeys(thread_vars, "1"))
    try:
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(cpu_norm_worker, [(target, sub[0])] * cores)  # start workers, fit the target once each
            t0 = time.perf_counter()
            normed = pool.map(cpu_norm_worker, [(target, p) for p in sub], chunksize=1)
            t_norm = time.perf_counter() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    torch.set_num_threads(cnn_threads)
    x = torch.from_numpy(np.stack(normed)).float().div(255).permute(0, 3, 1, 2).contiguous()
    model_cpu.eval()
    with torch.inference_mode():
        model_cpu(x)  # warm-up at the timed shape
        t0 = time.perf_counter()
        model_cpu(x)
        t_cnn = time.perf_counter() - t0
    return {
        "value": round(sample / (t_norm + t_cnn), 3), "unit": "patches/s", "cores": cores, "kind": "port",
        "sample": (f"{sample} of the workload's patches: oracle (NumPy restatement of the reference) Macenko "
                   f"transform over a {cores}-process pool ({sample / t_norm:.1f} patches/s) + torch-CPU fp32 "
                   f"resnet18 on {cnn_threads} threads ({sample / t_cnn:.1f} patches/s)"),
    }


def pmc_traffic(kernel_substr: str) -> dict | None:
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 --pmc passes of this round
    (``scripts/final_profile.sh``: FETCH_SIZE and WRITE_SIZE in separate runs, same 4096 x 224^2 workload).
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes actually read -- calibrated here
    on the apply kernels, whose reads are known: 2 x 301.6 MB = the 616.6 MB input -- WRITE_SIZE is taken as is."""
    import re

    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted((ROOT / "profiles").glob(f"*_stain_pmc_{counter}.txt"))
        if not files:
            return None
        for line in files[-1].read_text().splitlines():
            m = re.search(rf"{counter} mean=\s*([0-9.]+) n=\s*\d+\s+(.*)", line)
            if m and kernel_substr in m.group(2):
                vals[counter] = float(m.group(1)) * 1024.0
                vals["file_" + counter] = files[-1].name
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return {"bytes": 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"],
            "source": f"profiles/{vals['file_FETCH_SIZE']} (x2, gfx950) + profiles/{vals['file_WRITE_SIZE']}"}


def main() -> None:
    args = parse()
    import numpy as np
    import torch

    from tiatoolbox_amd import _lib, distributed as tdist
    from tiatoolbox_amd.models.architecture import get_pretrained_model
    from tiatoolbox_amd.tools import _stain_device as dev
    from tiatoolbox_amd.tools.stainnorm import get_normalizer
    from tiatoolbox_amd.utils import synth

    rank, world_size, local_rank = tdist.init_from_env()
    if world_size != args.gpus:
        if args.gpus != 1 or world_size != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MIOpen's solver search (done once per convolution shape during warm-up): 43 -> 36 ms for resnet18 fp16
    torch.backends.cudnn.benchmark = os.environ.get("TIA_MIOPEN_FIND", "1") == "1"
    dtype = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}[args.dtype]
    n, hw = args.patches, args.patch_size

    # ---- synthetic workload, resident in HBM ---------------------------------------------------
    uniq = min(n, 256)
    host = synth.g_he(uniq, hw, hw, seed=1 + rank)
    x = torch.from_numpy(host).to(device).repeat((n + uniq - 1) // uniq, 1, 1, 1)[:n].contiguous()
    target = np.load(ROOT / "tests" / "golden" / "target_crop_256.npy")
    norm = get_normalizer("macenko")
    norm.precision = args.precision
    norm.fit(target)
    import logging

    logging.getLogger("tiatoolbox_amd").setLevel(logging.ERROR)
    model, _ = get_pretrained_model("resnet18-kather100k")
    from tiatoolbox_amd.models.architecture.fused import fuse_cnn_model

    # eval copy: BatchNorm folded into the convolutions (MIOpen), bias/residual/ReLU/max-pool epilogues in HIP
    model_dev = fuse_cnn_model(model, epilogue_fusion="hip").to(device)
    if dtype != torch.float32:
        model_dev = model_dev.to(dtype)
    model_dev = model_dev.to(memory_format=torch.channels_last).eval()
    params = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
    out_kind = {torch.float16: _lib.OUT_UNIT_F16, torch.bfloat16: _lib.OUT_UNIT_BF16,
                torch.float32: _lib.OUT_UNIT_F32}[dtype]
    math = _lib.MATH_F32 if args.precision == "f32" else _lib.MATH_F64
    unit = torch.empty((n, hw, hw, 3), dtype=dtype, device=device)

    # library set-up outside any step: one forward per distinct micro-batch shape lets MIOpen finish its solver search
    # (seconds per convolution shape) before the first warm-up / timed step, whatever --warmup is
    unit.zero_()
    with torch.inference_mode():
        for m in sorted({min(args.micro_batch, n), n % args.micro_batch} - {0}):
            model_dev(unit[:m].permute(0, 3, 1, 2))
    torch.cuda.synchronize()

    def step() -> torch.Tensor:
        stats = dev.stain_stats(x, params)
        dev.stain_apply(x, stats, norm.stain_matrix_target, out_kind=out_kind, math=math, out=unit)
        probs = []
        with torch.inference_mode():
            for s in range(0, n, args.micro_batch):
                probs.append(model_dev(unit[s:s + args.micro_batch].permute(0, 3, 1, 2)))
            p = torch.cat(probs)
            pred = torch.argmax(p, dim=-1)
        if world_size > 1:
            p = tdist.all_gather_rows(p, n * world_size)
        return p, pred, stats

    def barrier() -> None:
        if world_size > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        p, pred, stats = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        p, pred, stats = step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world_size > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    dev.raise_on_flags(stats)
    assert p.shape == (n * world_size, 9) and bool(torch.isfinite(p).all())

    if rank != 0:
        return
    # ---- per-kernel timing with HIP events on the launch stream (kernels run on torch's current stream)
    def ev_time(fn, reps: int = 10) -> float:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    t_stats = ev_time(lambda: dev.stain_stats(x, params))
    t_apply = ev_time(lambda: dev.stain_apply(x, stats, norm.stain_matrix_target, out_kind=out_kind,
                                              math=math, out=unit))

    def cnn():
        with torch.inference_mode():
            for s in range(0, n, args.micro_batch):
                model_dev(unit[s:s + args.micro_batch].permute(0, 3, 1, 2))

    t_cnn = ev_time(cnn, reps=3)
    from tiatoolbox_amd.models.architecture.fused import hip_bias_act_

    act = torch.randn((args.micro_batch, 64, hw // 4, hw // 4), device=device).to(dtype).contiguous(memory_format=torch.channels_last)
    res = torch.randn_like(act)
    bias = torch.randn(64, device=device).to(dtype)
    t_epi = ev_time(lambda: hip_bias_act_(act, bias, res))
    px = n * hw * hw
    kernels = {
        # algorithmic bytes: stats reads the patch once (H*W*3 B); apply reads u8 + writes the CNN input
        "stain_stats_kernel": {"bound": "hbm", "seconds": t_stats, "alg_bytes": px * 3},
        "stain_apply_kernel": {"bound": "hbm", "seconds": t_apply, "alg_bytes": px * 3 * (1 + unit.element_size())},
    }
    kernels["bias_act_kernel(layer1, +residual)"] = {"bound": "hbm", "seconds": t_epi,
                                                     "alg_bytes": act.numel() * act.element_size() * 3}
    for k in kernels.values():
        k["achieved_GBs"] = k["alg_bytes"] / k["seconds"] / 1e9
        k["frac"] = k["achieved_GBs"] / HBM_PEAK_GBS
    dominant = max(kernels, key=lambda k: kernels[k]["seconds"])
    dk = kernels[dominant]
    flops = RESNET18_GFLOP_224 * (hw / 224.0) ** 2 * 1e9 * n
    roofline = {
        "kernel": dominant, "bound": "hbm", "achieved": round(dk["achieved_GBs"], 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(dk["frac"], 5), "traffic": None,
        "algorithmic_bytes": dk["alg_bytes"], "launch_ms": round(dk["seconds"] * 1e3, 4),
        "other_kernels": {
            name: {"bound": "hbm", "achieved": round(k["achieved_GBs"], 2), "unit": "GB/s",
                   "frac": round(k["frac"], 5), "launch_ms": round(k["seconds"] * 1e3, 4)}
            for name, k in kernels.items() if name != dominant},
        "backbone": {"bound": "mfma", "what": "resnet18 forward: MIOpen convolutions + hand-written HIP epilogues",
                     "achieved": round(flops / t_cnn / 1e12, 2), "peak": MFMA_PEAK_TFLOPS[args.dtype],
                     "unit": "TFLOP/s", "frac": round(flops / t_cnn / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 5),
                     "ms": round(t_cnn * 1e3, 3)},
    }
    pmc = pmc_traffic(dominant.split("(")[0]) if (n, hw) == (4096, 224) else None
    if pmc is not None:  # PMC passes cannot run inside the timed process; they are this round's committed profile
        roofline["traffic"] = round(pmc["bytes"])
        roofline["traffic_source"] = pmc["source"]
    total = n * world_size * args.steps
    line = {
        "metric": "patches/s, Macenko stain-norm + resnet18 PatchPredictor (synthetic patch batches)",
        "value": round(total / elapsed, 2), "unit": "patches/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (f"BASELINE configs[1]: PatchPredictor(resnet18-kather100k, seeded random weights) on "
                                f"{n} synthetic {hw}x{hw}x3 uint8 patches per GPU, Macenko pre-norm "
                                f"(stats f64, per-pixel {args.precision})"),
                   "patches_per_gpu": n, "patch_size": hw, "cnn_micro_batch": args.micro_batch,
                   "parallelism": f"dp{world_size} (patch-sharded, all_gather of probabilities)"},
        "roofline": roofline,
    }
    if not args.no_cpu_baseline and world_size == 1:
        cpu_model, _ = get_pretrained_model("resnet18-kather100k")
        line["cpu_baseline"] = cpu_baseline(target, host, cpu_model, args.cpu_sample)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
    try:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
