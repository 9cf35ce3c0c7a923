#!/usr/bin/env python
"""Benchmarks of the tiatoolbox per-patch hot path on MI355X.  Prints ONE JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W            # headline (BASELINE configs[1])
    python bench.py --config semantic|hovernet|vahadane ...   # BASELINE configs[2], [3], [4] (see bench_configs.py)

``--gpus N`` with N > 1 started plainly re-launches itself as ``python -m torch.distributed.run --nproc-per-node N``
(one rank per GPU, RCCL); started under ``torch.distributed.run`` it reads RANK / WORLD_SIZE / LOCAL_RANK as usual.

Headline workload = the configuration BASELINE.json's ``metric`` is quoted on: ``PatchPredictor("resnet18-kather100k")``
on 4096 synthetic **256x256x3** uint8 patches per GPU (G-he, seeded; seeded random weights -- pretrained weights are
unreachable offline), Macenko pre-normalisation fitted on the reference's target image crop (BASELINE ``configs[1]`` is
the same call on 224x224 patches: reported under ``extras.patch_224``).  One *step* = one call of the API the metric names,

    PatchPredictor.run(patches, patch_mode=True, return_probabilities=True, stain_normalizer=macenko)

over the rank's resident patches: per-patch Macenko statistics (HIP, f64), the normalising apply kernel (HIP, the
reference's f64 per-pixel arithmetic by default, uint8 out like the reference), resnet18 forward in **float32** on
hand-written kernels only (stem: uint8 in, ``ToTensor``'s 1/255 on load, conv7x7 + bias + ReLU + max-pool; blocks: MFMA
implicit GEMM -- the reference's arithmetic, ``vanilla.py:242``), softmax, argmax, the RCCL all-gather of the probabilities
(N > 1) and the copy of the result dict to host NumPy.  For ``value`` the uint8 patches are already resident in HBM when the timed region starts
(the engine's torch-tensor overload); the same call on HOST NumPy patches (H2D over PCIe included) is timed right after
and reported as ``host_inclusive`` (engine batches of 1024 there, so that copies hide behind compute; the resident run takes its
4096 patches as ONE engine batch: ``--micro-batch``).  ``value`` runs the engine's default ``conv_algo="auto"`` (float32 Winograd F(2x2, 3x3)
for the 3x3 / stride-1 block convolutions -- float32 in, float32 accumulate, gated by a committed per-layer error-bound test; executed and
effective flops reported separately).  Extras on rank 0 at N=1: ``cnn_direct`` = the same call with ``conv_algo="direct"`` (the audit mode:
direct implicit GEMM everywhere) with its kernels' rooflines, the fp16 backbone with its measured max |dp| against the
fp32 probabilities of the same batch (tolerance 1e-3, ``tests/engines/test_patch_predictor.py:719`` of the reference),
the 224x224 patch size of BASELINE configs[1], and -- ``extras.configs`` -- a SHORT run of each of BASELINE configs[2]-[4]
(``bench_configs.py``: semantic / hovernet / vahadane; value, step time, the config's roofline entry, a one-line CPU baseline), so
that the driver's default ``python bench.py`` observes all five configurations (``--no-extras`` skips them).  With N > 1 the line also carries every rank's own step time and the
time of the all-gather alone (``per_rank``), so that a scaling run explains itself.
"""

from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = {"float16": 2500.0, "bfloat16": 2500.0, "float32": 157.3}
RESNET18_GFLOP_224 = 3.64  # 1.82 GMAC per 224x224 patch; scales with the pixel count (4.75 at 256x256)


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="patch", choices=["patch", "semantic", "hovernet", "vahadane"])
    ap.add_argument("--patches", type=int, default=4096, help="patches per GPU per step")
    ap.add_argument("--patch-size", type=int, default=256, help="BASELINE.json metric: 256 (configs[1]: 224)")
    ap.add_argument("--micro-batch", type=int, default=4096, help="engine batch_size (CNN forward batch)")
    ap.add_argument("--dtype", default=os.environ.get("TIA_BENCH_DTYPE", "float32"),
                    choices=["float32", "float16", "bfloat16"], help="CNN arithmetic (reference: float32)")
    ap.add_argument("--precision", default="f64", choices=["f32", "f64"],
                    help="per-pixel arithmetic of the stain apply kernel (reference: f64; statistics are always f64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip host-inclusive / fp16 / 224^2 extra measurements and the short runs of BASELINE configs[2]-[4]")
    ap.add_argument("--cpu-sample", type=int, default=64)
    ap.add_argument("--slide", type=int, default=20000, help="--config semantic: slide edge in pixels")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (NumPy restatement of the reference) + torch-CPU fp32 resnet18
# ----------------------------------------------------------------------------------------------------------------------
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
    """Oracle Macenko + torch-CPU fp32 resnet18 on a bounded sample of the same workload: all host cores (process pool
    over patches, one BLAS thread each; CNN on <=64 threads) and ONE core; every figure is the median of 3 repeats."""
    import multiprocessing as mp

    import numpy as np
    import torch

    cores = os.cpu_count() or 1
    sample = min(max(sample, 2 * min(cores, 256)), len(patches))
    cnn_threads = min(cores, 64)
    sub = [np.ascontiguousarray(p) for p in patches[:sample]]
    thread_vars = ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in thread_vars}
    os.environ.update(dict.fromkeys(thread_vars, "1"))
    t_norm_all = []
    try:
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(cpu_norm_worker, [(target, sub[0])] * cores)  # start workers, fit the target once each
            for _ in range(3):
                t0 = time.perf_counter()
                normed = pool.map(cpu_norm_worker, [(target, p) for p in sub], chunksize=1)
                t_norm_all.append(time.perf_counter() - t0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    t_norm = statistics.median(t_norm_all)
    x = torch.from_numpy(np.stack(normed)).float().div(255).permute(0, 3, 1, 2).contiguous()
    model_cpu.eval()

    def cnn_time(inp, threads: int) -> float:
        torch.set_num_threads(threads)
        ts = []
        with torch.inference_mode():
            model_cpu(inp)  # warm-up at the timed shape
            for _ in range(3):
                t0 = time.perf_counter()
                model_cpu(inp)
                ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    t_cnn = cnn_time(x, cnn_threads)
    # one core: a handful of patches through the same two stages, in this process
    one = sub[:4]
    ts = []
    cpu_norm_worker((target, one[0]))
    for _ in range(3):
        t0 = time.perf_counter()
        for p in one:
            cpu_norm_worker((target, p))
        ts.append(time.perf_counter() - t0)
    t_norm1 = statistics.median(ts) / len(one)
    t_cnn1 = cnn_time(x[:4], 1) / 4
    torch.set_num_threads(cnn_threads)
    return {
        "value": round(sample / (t_norm + t_cnn), 3), "unit": "patches/s", "cores": cores, "kind": "port",
        "repeats": 3, "statistic": "median",
        "one_core": {"value": round(1.0 / (t_norm1 + t_cnn1), 3), "unit": "patches/s", "cores": 1,
                     "macenko_patches_per_s": round(1.0 / t_norm1, 3), "resnet18_patches_per_s": round(1.0 / t_cnn1, 3),
                     "sample": "4 patches, single thread"},
        "sample": (f"{sample} of the workload's patches: oracle (NumPy restatement of the reference) Macenko "
                   f"transform over a {cores}-process pool ({sample / t_norm:.1f} patches/s) + torch-CPU fp32 "
                   f"resnet18 on {cnn_threads} threads ({sample / t_cnn:.1f} patches/s)"),
    }


def pmc_traffic(kernel_substr: str, stem: str, calls_per_forward: int | None = None) -> dict | None:
    """HBM-side bytes per launch of a kernel from this round's committed rocprofv3 --pmc passes
    (``profiles/*_<stem>_pmc_FETCH_SIZE.txt`` / ``..._WRITE_SIZE.txt``: separate runs over the same workload; mean over the
    dispatches of every kernel whose name contains ``kernel_substr``, weighted by dispatch count).  A layer call on a 4096-patch
    batch is several dispatches (inputs go in groups of < 2 GiB), so when the pass says how many forwards it ran (``# PMC
    forwards=N``, ``scripts/perf_trunk.py ... pmc``) and the caller knows the kernel's calls per forward, the figure is bytes per
    CALL = counter total / (N x calls_per_forward) -- the unit ``launch_ms`` and ``algorithmic_flops`` are in.
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes actually read -- calibrated here on the
    apply kernels, whose reads are known (2 x 301.6 MB = the 616.6 MB input); WRITE_SIZE is taken as is.  Units: KiB."""
    import re

    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted((ROOT / "profiles").glob(f"*_{stem}_pmc_{counter}.txt"))
        if not files:
            return None
        tot, cnt, forwards = 0.0, 0, None
        for line in files[-1].read_text().splitlines():
            m = re.search(rf"{counter} mean=\s*([0-9.]+) n=\s*(\d+)\s+(.*)", line)
            if m and kernel_substr in m.group(3):
                tot += float(m.group(1)) * int(m.group(2))
                cnt += int(m.group(2))
            f = re.match(r"# PMC forwards=(\d+)", line)
            if f:
                forwards = int(f.group(1))
        if cnt == 0:
            return None
        per_call = bool(forwards and calls_per_forward)
        vals[counter] = tot / (forwards * calls_per_forward if per_call else cnt) * 1024.0
        vals["file_" + counter] = files[-1].name
        vals["unit"] = (f"per layer call: total over {cnt} dispatches / ({forwards} forwards x {calls_per_forward} calls)" if per_call
                        else "mean per dispatch")
    return {"bytes": 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"],
            "source": (f"profiles/{vals['file_FETCH_SIZE']} (x2, gfx950) + profiles/{vals['file_WRITE_SIZE']}, {vals['unit']}"
                       + _pass_commit(vals["file_FETCH_SIZE"]))}


def pmc_mfma_busy(stem: str, kernel_substr: str) -> dict | None:
    """Fraction of the run time the matrix pipes of the 1024 SIMDs were busy under a kernel, from this round's committed counter pass
    (``profiles/*_<stem>_pmc_MFMA.txt``, ``scripts/pmc_mfma.sh``): ``SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)``, dispatch-
    weighted over every kernel whose name contains ``kernel_substr`` (both counters come from the SAME pass)."""
    import re

    files = sorted((ROOT / "profiles").glob(f"*_{stem}_pmc_MFMA.txt"))
    note = " (the busy counter advances in steps of 2^28 cycles per dispatch: +-3 % on these launches)"
    if not files:
        return None
    busy = act = 0.0
    for line in files[-1].read_text().splitlines():
        m = re.search(r"(SQ_VALU_MFMA_BUSY_CYCLES|GRBM_GUI_ACTIVE) mean=\s*([0-9.]+) n=\s*(\d+)\s+(.*)", line)
        if m and kernel_substr in m.group(4):
            if m.group(1) == "GRBM_GUI_ACTIVE":
                act += float(m.group(2)) * int(m.group(3))
            else:
                busy += float(m.group(2)) * int(m.group(3))
    if act == 0.0:
        return None
    return {"mfma_busy": round(busy / (1024.0 * act / 8.0), 4),
            "source": f"profiles/{files[-1].name}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs){note}{_pass_commit(files[-1].name)}"}


def _pass_commit(profile_name: str) -> str:
    """The commit whose code a committed counter pass measured (``profiles/<tag>_COMMIT.txt``, written when the pass is copied in)."""
    note = ROOT / "profiles" / (profile_name.split("_")[0] + "_COMMIT.txt")
    return f"; code at commit {note.read_text().strip()}" if note.exists() else ""


def pmc_traffic_per_call(stem: str, kernel_substrs: tuple[str, ...], per_call_kernel: str) -> dict | None:
    """HBM-side bytes per CALL of an operation that is several launches (``scripts/pmc_workloads.py <stem>``): the counter totals of
    every dispatch whose kernel name contains one of ``kernel_substrs``, divided by the number of calls = the dispatch count of
    ``per_call_kernel`` (a kernel the operation launches exactly once).  Same files, units and gfx950 correction as ``pmc_traffic``."""
    import re

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
            if any(k in m.group(3) for k in kernel_substrs):
                tot += float(m.group(1)) * int(m.group(2))
            if per_call_kernel in m.group(3):
                calls += int(m.group(2))
        if calls == 0:
            return None
        vals[counter] = tot / calls * 1024.0
        vals["file_" + counter] = files[-1].name
    return {"bytes": 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"],
            "source": f"profiles/{vals['file_FETCH_SIZE']} (x2, gfx950) + profiles/{vals['file_WRITE_SIZE']}{_pass_commit(vals['file_FETCH_SIZE'])}; per call = totals over "
                      f"{list(kernel_substrs)} / dispatches of {per_call_kernel}"}


def ev_time(fn, reps: int = 10) -> float:
    """Average seconds per call from HIP events on the launch stream (kernels run on torch's current stream)."""
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


def _lib_route(n, h, w, cin, cout, k, stride, pad, ho, wo) -> int:
    from tiatoolbox_amd import _lib

    route = int(_lib.load().tia_conv2d_route_f32(n, h, w, cin, cout, k, k, stride, pad, pad, ho, wo))
    if route < 0:
        raise RuntimeError(f"tia_conv2d_route_f32 refused the shape ({route})")
    return route


def trunk_roofline(model, u8_batch):
    """HIP-event times and flops of the hand-written convolution kernels of one trunk forward: the stem kernel (one launch) and the
    block convolutions, split by kernel -- ``conv3x3_wino_kernel`` (3x3 / stride 1 through Winograd F(2x2, 3x3): the engine's default,
    ``conv_algo="auto"``), ``conv3x3_spatial_kernel`` (the same layers in ``conv_algo="direct"``: tap reuse), ``conv1x1_ring_kernel``
    (1x1, and strided 3x3 gathered) and ``conv_mfma_f32_kernel`` (what is left), as ``tia_conv2d_route_f32`` says.  Per-launch events
    (one sync per launch, kernel time only) give the split; the total is timed separately over whole forwards without syncs.
    For the Winograd kernel ``flops`` are the DIRECT convolution's 2*M*Cout*Cin*9 (what the layer computes) and ``exec_flops`` what its
    MFMAs execute -- 16 multiplies per 2 x 2 output tile instead of 36, counted over the 64-tile x 64-channel blocks actually launched
    (tiles beyond the map included)."""
    import torch

    import tiatoolbox_amd.models.architecture.fused as fused
    from tiatoolbox_amd.models.architecture.fused import MfmaResNet

    trunk = next((m for m in model.modules() if isinstance(m, MfmaResNet)), None)
    if trunk is None:
        return None
    names = ("conv3x3_wino_kernel", "conv3x3_spatial_kernel", "conv1x1_ring_kernel", "conv_mfma_f32_kernel")
    fam = {k: [0, 0.0, 0, 0] for k in names}  # launches, s, algorithmic flops, executed flops
    plain, plain_w = fused.hip_conv2d, fused.hip_conv3x3_wino

    def timed(x, w, b, residual, *, kernel, stride, padding, relu):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = plain(x, w, b, residual, kernel=kernel, stride=stride, padding=padding, relu=relu)
        e1.record()
        e1.synchronize()
        n, co, ho, wo = y.shape
        route = _lib_route(n, x.shape[2], x.shape[3], x.shape[1], co, kernel, stride, padding, ho, wo)
        name = ("conv_mfma_f32_kernel", "conv3x3_spatial_kernel", "conv1x1_ring_kernel")[route]  # the library's own answer
        f = fam[name]
        f[0] += 1
        f[1] += e0.elapsed_time(e1) * 1e-3
        f[2] += 2 * n * ho * wo * co * x.shape[1] * kernel * kernel
        f[3] += 2 * n * ho * wo * co * x.shape[1] * kernel * kernel
        return y

    def timed_w(x, u, b, residual, *, padding, relu, pad_hi=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = plain_w(x, u, b, residual, padding=padding, relu=relu, pad_hi=pad_hi)
        e1.record()
        e1.synchronize()
        n, co, ho, wo = y.shape
        cin = x.shape[1]
        small = ho <= 8 and wo <= 8
        blocks = -(-n // 4) if small else n * (-(-ho // 16)) * (-(-wo // 16))  # 64 tiles each
        f = fam["conv3x3_wino_kernel"]
        f[0] += 1
        f[1] += e0.elapsed_time(e1) * 1e-3
        f[2] += 2 * n * ho * wo * co * cin * 9
        f[3] += 2 * blocks * 64 * 16 * cin * co
        return y

    reps = 3
    with torch.inference_mode():
        feat = trunk.stem_forward(u8_batch)
        nb, hin, win = u8_batch.shape[0], u8_batch.shape[1], u8_batch.shape[2]
        ho, wo = (hin - 1) // 2 + 1, (win - 1) // 2 + 1
        stem_flops = 2 * nb * ho * wo * 64 * 147
        trunk.blocks(feat)
        fused.hip_conv2d, fused.hip_conv3x3_wino = timed, timed_w
        try:
            for _ in range(reps):
                trunk.blocks(feat)
        finally:
            fused.hip_conv2d, fused.hip_conv3x3_wino = plain, plain_w
        seconds = ev_time(lambda: trunk.blocks(feat), reps=5)
        stem_seconds = ev_time(lambda: trunk.stem_forward(u8_batch), reps=5)
    launches = sum(f[0] for f in fam.values()) // reps
    flops = sum(f[2] for f in fam.values()) // reps
    exec_flops = sum(f[3] for f in fam.values()) // reps
    kernels = {name: {"launches": f[0] // reps, "seconds": f[1] / reps, "flops": f[2] // reps, "exec_flops": f[3] // reps,
                      "tflops": f[3] / f[1] / 1e12 if f[1] > 0 else 0.0, "effective_tflops": f[2] / f[1] / 1e12 if f[1] > 0 else 0.0}
               for name, f in fam.items() if f[0]}
    return {"seconds": seconds, "launches": launches, "flops_per_launch": flops // launches,
            "ms_per_launch": seconds / launches * 1e3, "tflops": flops / seconds / 1e12, "exec_tflops": exec_flops / seconds / 1e12,
            "flops": flops, "exec_flops": exec_flops, "kernels": kernels,
            "stem_seconds": stem_seconds, "stem_flops": stem_flops, "stem_tflops": stem_flops / stem_seconds / 1e12}


def _condense(full: dict) -> dict:
    """The part of a ``bench_configs`` line that ``extras.configs`` carries: value, step time, the roofline of the config's own
    dominant kernel (+ the network's forward), and a one-line CPU baseline."""
    roof = full.get("roofline", {})
    out = {"metric": full["metric"], "value": full["value"], "unit": full["unit"], "ms_per_step": full["ms_per_step"],
           "steps": full["steps"], "warmup": full["warmup"], "dtype": full["dtype"], "scaling": full["scaling"],
           "workload": full["config"]["workload"],
           "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms",
                                                 "algorithmic_bytes") if k in roof}}
    if "backbone" in roof:
        out["roofline"]["backbone"] = {k: roof["backbone"].get(k) for k in ("achieved", "peak", "unit", "frac")}
    cpu = full.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cpu}
    for key in ("cnn_direct", "cnn_winograd"):
        alt = (full.get("extras") or {}).get(key)
        if alt:
            out[key] = {k: v for k, v in alt.items() if k != "note"}
    return out


def config_extras(args: argparse.Namespace) -> dict:
    """BASELINE configs[2]-[4] inside the default run, so the driver's bench record observes them too: a SHORT run of each
    ``bench_configs`` function (same code as ``bench.py --config ...``; fewer steps), condensed.  A config that fails is
    reported as ``{"error": ...}`` and does not take the headline down with it."""
    import copy
    import gc

    import torch

    import bench_configs

    plan = {"vahadane": (3, 1), "hovernet": (2, 1), "semantic": (1, 1)}  # (timed steps, warm-up steps)
    out = {}
    for name, (steps, warmup) in plan.items():
        a = copy.copy(args)
        a.config, a.steps, a.warmup, a.patches = name, steps, warmup, 4096  # 4096 = "the config's own default size"
        t0 = time.perf_counter()
        try:
            out[name] = _condense(getattr(bench_configs, f"bench_{name}")(a))
        except Exception as exc:  # noqa: BLE001
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        out[name]["bench_wall_s"] = round(time.perf_counter() - t0, 1)
        gc.collect()
        torch.cuda.empty_cache()
    return out


def self_spawn(args: argparse.Namespace) -> None:
    """``python bench.py --gpus N`` outside torchrun: become ``torch.distributed.run`` with N local ranks."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvp(sys.executable, cmd)  # noqa: S606


def verify_ranks(args: argparse.Namespace, world_size: int, local_rank: int) -> dict | None:
    """Fail loudly instead of reporting an N-GPU number that was not measured on N GPUs: the process group must have
    ``--gpus`` ranks, and no two ranks may sit on the same device (same host + device index, or same device UUID).  Returns what
    the line reports as ``rccl`` for N > 1: world size, backend, every rank's host / device index / device UUID."""
    import socket

    import torch
    import torch.distributed as dist

    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world_size} rank(s)")
    if world_size == 1:
        return None
    if not dist.is_initialized() or dist.get_world_size() != args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: torch.distributed is not initialised with that many ranks")
    if dist.get_backend() != "nccl":
        raise SystemExit(f"--gpus {args.gpus}: backend {dist.get_backend()!r}, expected 'nccl' (RCCL)")
    if torch.cuda.device_count() < 1 or local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank with LOCAL_RANK={local_rank} sees {torch.cuda.device_count()} device(s)")
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    mine = (socket.gethostname(), torch.cuda.current_device(), str(getattr(props, "uuid", "")))
    seen: list = [None] * world_size
    dist.all_gather_object(seen, mine)
    if len({(h, d) for h, d, _ in seen}) != world_size or (all(u for _, _, u in seen) and len({u for _, _, u in seen}) != world_size):
        raise SystemExit(f"--gpus {args.gpus}: ranks share a device: {seen}")
    return {"world_size": world_size, "backend": dist.get_backend(), "hosts": [h for h, _, _ in seen],
            "device_indices": [d for _, d, _ in seen], "device_uuids": [u for _, _, u in seen],
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}


def bench_patch(args: argparse.Namespace) -> dict | None:
    import logging

    import numpy as np
    import torch

    from tiatoolbox_amd import _lib, distributed as tdist
    from tiatoolbox_amd.models.architecture import get_pretrained_model
    from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor
    from tiatoolbox_amd.tools import _stain_device as dev
    from tiatoolbox_amd.tools.stainnorm import get_normalizer
    from tiatoolbox_amd.utils import synth

    rank, world_size, local_rank = tdist.init_from_env()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rccl_info = verify_ranks(args, world_size, local_rank)
    logging.getLogger("tiatoolbox_amd").setLevel(logging.ERROR)
    n, hw = args.patches, args.patch_size
    target = np.load(ROOT / "tests" / "golden" / "target_crop_256.npy")
    norm = get_normalizer("macenko")
    norm.precision = args.precision
    norm.fit(target)

    def workload(size: int, n_total: int) -> tuple[np.ndarray, torch.Tensor]:
        """The job's patches: 256 unique G-he patches (seeded), repeated -- every rank holds the whole list, the
        engine takes its contiguous shard; weak scaling: ``n`` patches per GPU."""
        uniq = min(n_total, 256)
        host = synth.g_he(uniq, size, size, seed=1)
        x = torch.from_numpy(host).to(device).repeat((n_total + uniq - 1) // uniq, 1, 1, 1)[:n_total].contiguous()
        return host, x

    host, x = workload(hw, n * world_size)
    engine = PatchPredictor(model="resnet18-kather100k", batch_size=args.micro_batch, device=f"cuda:{local_rank}",
                            verbose=False)

    def run(images, dtype: str = args.dtype, algo: str = "auto"):
        """The API call the metric names; ``conv_algo`` is passed explicitly (run kwargs persist on the engine): "auto" is the
        engine's default (Winograd F(2x2, 3x3) for the float32 3x3 / stride-1 block convolutions), "direct" the audit mode."""
        size = tuple(int(v) for v in images.shape[1:3])
        return engine.run(images, patch_mode=True, return_probabilities=True, stain_normalizer=norm,
                          patch_input_shape=size, compute_dtype=dtype, conv_algo=algo)

    def barrier() -> None:
        if world_size > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(images, steps: int, warmup: int, dtype: str = args.dtype, algo: str = "auto"):
        out = None
        for _ in range(warmup):
            out = run(images, dtype, algo)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run(images, dtype, algo)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0  # this rank's own time, before it waits for the others
        barrier()
        return time.perf_counter() - t0, out, own

    run(x)  # lazy set-up (module loads, weight packing, workspace growth) outside any step, whatever --warmup is
    elapsed, out, own = timed(x, args.steps, args.warmup)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    per_rank = None
    if world_size > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        # diagnostics for the scaling run: every rank's own step time, and the all-gather of the probabilities alone
        owns = torch.zeros(world_size, dtype=torch.float64, device=device)
        owns[rank] = own / args.steps * 1e3
        torch.distributed.all_reduce(owns)
        local = torch.rand((n, 9), device=device)
        tdist.all_gather_rows(local, n * world_size)
        barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            tdist.all_gather_rows(local, n * world_size)
        barrier()
        per_rank = {"own_ms_per_step": [round(float(v), 3) for v in owns.cpu().tolist()],
                    "allgather_probabilities_ms": round((time.perf_counter() - t0) / 20 * 1e3, 4),
                    "allgather_bytes_per_rank": n * 9 * 4,
                    "what": ("own = rank-local wall time per step before the closing barrier (the step already contains the "
                             "all-gather, which synchronises the ranks); allgather = 20 back-to-back padded "
                             "all_gather_into_tensor calls of the [patches_per_gpu, 9] float32 probabilities")}
    elapsed = float(t.item())
    probs = out["probabilities"]
    assert probs.shape == (n * world_size, 9) and np.isfinite(probs).all()
    assert out["predictions"].shape == (n * world_size,)
    if rank != 0:
        return None

    total = n * world_size * args.steps
    line = {
        "metric": "patches/s (256x256x3), Macenko stain-norm + resnet18 PatchPredictor.run() (synthetic patch batches)",
        "value": round(total / elapsed, 2), "unit": "patches/s", "n_gpus": world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (f"{n} x {hw}x{hw}x3 u8 patches/GPU in HBM, PatchPredictor(resnet18-kather100k).run(), Macenko pre-norm "
                                f"(f64 stats, {args.precision} apply), CNN {args.dtype} hand-written HIP, conv_algo=auto (Winograd F(2x2,3x3) "
                                f"on 3x3/s1; direct: extras.cnn_direct); seeded random weights; configs[1] 224^2: extras.patch_224"),
                   "api": "PatchPredictor.run(images, patch_mode=True, return_probabilities=True, stain_normalizer=...)",
                   "patches_per_gpu": n, "patch_size": hw, "engine_batch_size": args.micro_batch,
                   "parallelism": f"dp{world_size} (patch-sharded, all_gather of probabilities)"},
    }
    if world_size > 1:
        assert per_rank is not None and rccl_info is not None  # a multi-GPU line always explains itself
        line["per_rank"] = per_rank
        line["rccl"] = rccl_info

    # ---- per-kernel timing with HIP events on the launch stream ------------------------------------------------
    dtype_t = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}[args.dtype]
    xs = x[:n]
    mb = args.micro_batch
    params = norm.extractor.stats_params(target_stain=norm.stain_matrix_target, target_maxc=norm.maxC_target)
    # float32: the apply kernel writes the reference's uint8 result and the stem kernel reads it (ToTensor on load);
    # fp16 / bf16: the apply kernel writes the unit-scaled half tensor for the library convolutions
    out_kind = {torch.float16: _lib.OUT_UNIT_F16, torch.bfloat16: _lib.OUT_UNIT_BF16, torch.float32: _lib.OUT_U8}[dtype_t]
    math = _lib.MATH_F32 if args.precision == "f32" else _lib.MATH_F64
    unit = torch.empty((n, hw, hw, 3), dtype=torch.uint8 if dtype_t == torch.float32 else dtype_t, device=device)
    stats = dev.stain_stats(xs, params)
    t_stats = ev_time(lambda: dev.stain_stats(xs, params))
    t_apply = ev_time(lambda: dev.stain_apply(xs, stats, norm.stain_matrix_target, out_kind=out_kind, math=math,
                                              out=unit))
    model_dev = engine._inference_model(dtype_t)  # noqa: SLF001  (the engine's own BN-folded inference copy)

    def cnn():
        with torch.inference_mode(), engine._miopen_scope():  # noqa: SLF001
            for s in range(0, n, mb):
                model_dev(unit[s:s + mb].permute(0, 3, 1, 2))

    t_cnn = ev_time(cnn, reps=3)
    px = n * hw * hw
    kernels = {
        # algorithmic bytes: stats reads the patch once (H*W*3 B); apply reads u8 + writes the CNN input
        "stain_stats_kernel": {"seconds": t_stats, "alg_bytes": px * 3},
        "stain_apply_kernel": {"seconds": t_apply, "alg_bytes": px * 3 * (1 + unit.element_size())},
    }
    for k in kernels.values():
        k["achieved_GBs"] = k["alg_bytes"] / k["seconds"] / 1e9
        k["frac"] = k["achieved_GBs"] / HBM_PEAK_GBS
    flops = RESNET18_GFLOP_224 * (hw / 224.0) ** 2 * 1e9 * n
    other = {
        name: {"bound": "hbm", "achieved": round(k["achieved_GBs"], 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(k["frac"], 5), "launch_ms": round(k["seconds"] * 1e3, 4), "algorithmic_bytes": k["alg_bytes"]}
        for name, k in kernels.items()}
    steps_mb = (n + mb - 1) // mb
    conv = trunk_roofline(model_dev, unit[:mb]) if args.dtype == "float32" else None
    if conv is not None:
        # the hand-written kernel a step spends most of its time in (MFMA-bound): of the two block-convolution kernels the one
        # with the larger share of the trunk; the other one and the whole trunk are listed beside it
        fams = conv["kernels"]
        dominant = max(fams, key=lambda k: fams[k]["seconds"])
        dk = fams[dominant]
        desc = {"conv3x3_wino_kernel": ("3x3 / stride-1 convolutions through Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32 (float32 in / float32 "
                                        "accumulate, weights transformed once in float64); `achieved` / `frac` from the flops the MFMAs EXECUTE (16 "
                                        "multiplies per 2x2 outputs, 64-tile x 64-channel blocks as launched), `effective_tflops` = the direct "
                                        "convolution's 2*M*Cout*Cin*9 over the same time (may exceed the MFMA peak)"),
                "conv3x3_spatial_kernel": "3x3 / stride-1 convolutions, 16x16 (or 2 x 8x8) pixel blocks with tap reuse (LDS-DMA patch + weight ring)",
                "conv1x1_ring_kernel": "1x1 and strided 3x3 convolutions as a GEMM over 256-pixel blocks, both operands by LDS-DMA (two-stage ring, taps gathered)",
                "conv_mfma_f32_kernel": "convolutions left to the register-staged 128-pixel slice kernel (small maps / few workgroups)"}
        roofline = {
            "kernel": dominant, "bound": "mfma", "achieved": round(dk["tflops"], 2),
            "peak": MFMA_PEAK_TFLOPS["float32"], "unit": "TFLOP/s",
            "frac": round(dk["tflops"] / MFMA_PEAK_TFLOPS["float32"], 5), "traffic": None,
            "algorithmic_flops": dk["flops"] // dk["launches"], "executed_mfma_flops": dk["exec_flops"] // dk["launches"],
            "effective_tflops": round(dk["effective_tflops"], 2), "launch_ms": round(dk["seconds"] / dk["launches"] * 1e3, 4),
            "launches_per_step": dk["launches"] * steps_mb,
            "what": (f"average over the {dk['launches']} launches of this kernel in one resnet18 forward on {mb} patches of "
                     f"{hw}x{hw} ({desc[dominant]}; fp32 MFMA 32x32x2, bias + residual + ReLU fused); "
                     "per-launch HIP events on the launch stream"),
            "share_of_step": round(dk["seconds"] * steps_mb / (elapsed / args.steps), 3),
            "trunk": {"what": (f"all {conv['launches']} block convolutions of one forward, timed back to back; `achieved` / `frac` from the "
                               "flops the MFMAs execute, `effective_tflops` from the direct convolutions' count"),
                      "achieved": round(conv["exec_tflops"], 2), "frac": round(conv["exec_tflops"] / MFMA_PEAK_TFLOPS["float32"], 5),
                      "effective_tflops": round(conv["tflops"], 2), "ms": round(conv["seconds"] * 1e3, 3),
                      "share_of_step": round(conv["seconds"] * steps_mb / (elapsed / args.steps), 3)},
        }
        for name, k in fams.items():
            if name != dominant:
                other[name] = {"bound": "mfma", "achieved": round(k["tflops"], 2), "peak": MFMA_PEAK_TFLOPS["float32"],
                               "unit": "TFLOP/s", "frac": round(k["tflops"] / MFMA_PEAK_TFLOPS["float32"], 5),
                               "launch_ms": round(k["seconds"] / k["launches"] * 1e3, 4), "launches_per_step": k["launches"] * steps_mb,
                               "algorithmic_flops": k["flops"] // k["launches"], "what": desc[name]}
        other["stem7x7_pool_kernel"] = {
            "bound": "mfma", "achieved": round(conv["stem_tflops"], 2), "peak": MFMA_PEAK_TFLOPS["float32"],
            "unit": "TFLOP/s", "frac": round(conv["stem_tflops"] / MFMA_PEAK_TFLOPS["float32"], 5),
            "launch_ms": round(conv["stem_seconds"] * 1e3, 4), "algorithmic_flops": conv["stem_flops"],
            "what": (f"uint8 patches -> conv7x7/2 (3->64, K = 147) + bias + ReLU + maxpool3x3/2, one launch per {mb} patches; "
                     "flops = 2*Ho*Wo*64*147 per patch (the conv rows recomputed at chunk seams are not counted)")}
        # PMC passes cannot run inside the timed process: this round's committed passes over the same shapes (stem "trunk" = 1024-patch
        # launches, "trunk<mb>" otherwise)
        stem_name = ("wino" if dominant == "conv3x3_wino_kernel" else "trunk") + ("" if mb == 1024 else str(mb))
        pmc_c = pmc_traffic(dominant, stem_name, dk["launches"]) if hw == 256 else None
        busy = pmc_mfma_busy(stem_name, dominant) if hw == 256 else None
        if busy is not None:
            roofline["mfma_busy"] = busy["mfma_busy"]
            roofline["mfma_busy_source"] = busy["source"]
        if pmc_c is not None:
            roofline["traffic"] = round(pmc_c["bytes"])
            roofline["traffic_source"] = pmc_c["source"] + (f" (workload: scripts/perf_wino.py {mb} 256 pmc)" if dominant == "conv3x3_wino_kernel"
                                                             else f" (workload: scripts/perf_trunk.py {mb} 256 [5 pmc])")
    else:
        dominant = "stain_stats_kernel"  # the longest-running hand-written kernel of a step when the library convolves
        dk = kernels[dominant]
        roofline = {
            "kernel": dominant, "bound": "hbm", "achieved": round(dk["achieved_GBs"], 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(dk["frac"], 5), "traffic": None,
            "algorithmic_bytes": dk["alg_bytes"], "launch_ms": round(dk["seconds"] * 1e3, 4),
        }
        other.pop(dominant)
    roofline["other_kernels"] = other
    # flops the Winograd layers do not execute (per forward of the whole step's patches): `achieved` prices executed work only
    saved = (conv["flops"] - conv["exec_flops"]) * steps_mb if conv is not None else 0
    roofline["backbone"] = {
        "bound": "mfma", "what": ("resnet18 forward, every convolution hand-written (stem kernel + MFMA implicit GEMM with fused "
                                  "epilogues); average pool / classifier GEMM / softmax: torch" if conv is not None
                                  else "resnet18 forward: library convolutions + hand-written HIP epilogues"),
        "achieved": round((flops - saved) / t_cnn / 1e12, 2), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
        "frac": round((flops - saved) / t_cnn / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 5),
        "effective_tflops": round(flops / t_cnn / 1e12, 2), "ms": round(t_cnn * 1e3, 3)}
    # which of the two statistics kernels serves this shape (the library's own answer), and its HBM-side traffic per launch
    import ctypes

    reg = _lib.load().tia_stain_stats_path(hw, hw, ctypes.byref(params)) == 1
    stats_kernel = "stain_stats_reg_kernel" if reg else "stain_stats_kernel<false>"
    tgt = roofline if dominant == "stain_stats_kernel" else roofline["other_kernels"]["stain_stats_kernel"]
    tgt["what"] = (f"{stats_kernel}: " + ("one 1024-thread workgroup per patch holds it in registers (one HBM read); patches it hands "
                                         "back go through the streaming kernel" if reg else
                                         "one 512-thread workgroup per patch, the patch re-read per sweep"))
    pmc = pmc_traffic(stats_kernel.split("<")[0] + ("(" if reg else "<false>"), "stain") if (n, hw) == (4096, 256) else None
    if pmc is not None:
        tgt["traffic"] = round(pmc["bytes"])
        tgt["traffic_source"] = pmc["source"] + (" (scripts/perf_stain.py 4096 256; mean over the dispatches of that kernel)")
    line["roofline"] = roofline

    # ---- extras (rank 0, single GPU): host-inclusive API call, fp16 backbone with its error, 224^2 patches --------
    if world_size == 1 and not args.no_extras:
        extras = {}
        k_extra = max(3, min(args.steps, 5))
        host_all = np.ascontiguousarray(np.tile(host, ((n + len(host) - 1) // len(host), 1, 1, 1))[:n])
        # host data wants engine batches SMALLER than the step, so that the copy of batch i + 1 hides behind the compute of batch i
        host_mb = min(args.micro_batch, 1024)
        engine.batch_size = host_mb
        el, out_h, _ = timed(host_all, k_extra, 1)
        engine.batch_size = args.micro_batch
        assert float((out_h["predictions"] == out["predictions"][:n]).mean()) > 0.999
        line["host_inclusive"] = {
            "value": round(n * k_extra / el, 2), "unit": "patches/s", "ms_per_step": round(el / k_extra * 1e3, 3),
            "steps": k_extra, "engine_batch_size": host_mb,
            "what": ("the same PatchPredictor.run() call on HOST NumPy uint8 patches: page-lock in place + H2D over PCIe (one batch "
                     "ahead, copy stream) + compute + D2H of results"),
            "pcie_GBs": round(n * hw * hw * 3 * k_extra / el / 1e9, 2)}
        del host_all
        if args.dtype == "float32":
            run(xs, "float16")
            el16, out16, _ = timed(xs, k_extra, 1, "float16")
            dp = float(np.abs(out16["probabilities"].astype(np.float64) - probs[:n].astype(np.float64)).max())
            extras["cnn_float16"] = {
                "value": round(n * k_extra / el16, 2), "unit": "patches/s", "ms_per_step": round(el16 / k_extra * 1e3, 3),
                "max_abs_dprob_vs_float32": dp, "tolerance": 1e-3, "within_tolerance": bool(dp <= 1e-3),
                "argmax_agreement": float((out16["predictions"] == out["predictions"][:n]).mean()),
                "note": "extra only: fp16 backbone (fp32 accumulate), same batch; not the reported value"}
        if args.dtype == "float32":
            # the audit mode: every block convolution as a direct implicit GEMM (the reference's order of accumulation over taps);
            # the same call, the same batch -- its rate, the largest probability difference to the default path, and the direct
            # kernels' own rooflines
            run(xs, "float32", "direct")
            el_d, out_d, _ = timed(xs, k_extra, 1, "float32", "direct")
            dpd = float(np.abs(out_d["probabilities"].astype(np.float64) - probs[:n].astype(np.float64)).max())
            model_d = engine._inference_model(torch.float32)  # noqa: SLF001  (the direct copy: the engine's conv_algo is still "direct")
            conv_d = trunk_roofline(model_d, unit[:mb])
            engine.conv_algo = "auto"
            dkern = {}
            for name, k in (conv_d["kernels"] if conv_d else {}).items():
                dkern[name] = {"bound": "mfma", "achieved": round(k["tflops"], 2), "peak": MFMA_PEAK_TFLOPS["float32"], "unit": "TFLOP/s",
                               "frac": round(k["tflops"] / MFMA_PEAK_TFLOPS["float32"], 5), "launches_per_forward": k["launches"],
                               "launch_ms": round(k["seconds"] / k["launches"] * 1e3, 4), "algorithmic_flops": k["flops"] // k["launches"]}
            pmc_d = pmc_traffic("conv3x3_spatial_kernel", "trunk" if mb == 1024 else f"trunk{mb}", 13) if hw == 256 else None
            busy_d = pmc_mfma_busy("trunk" if mb == 1024 else f"trunk{mb}", "conv3x3_spatial_kernel") if hw == 256 else None
            if busy_d and "conv3x3_spatial_kernel" in dkern:
                dkern["conv3x3_spatial_kernel"]["mfma_busy"] = busy_d["mfma_busy"]
                dkern["conv3x3_spatial_kernel"]["mfma_busy_source"] = busy_d["source"]
            if pmc_d and "conv3x3_spatial_kernel" in dkern:
                dkern["conv3x3_spatial_kernel"]["traffic"] = round(pmc_d["bytes"])
                dkern["conv3x3_spatial_kernel"]["traffic_source"] = pmc_d["source"]
            extras["cnn_direct"] = {
                "value": round(n * k_extra / el_d, 2), "unit": "patches/s", "ms_per_step": round(el_d / k_extra * 1e3, 3),
                "max_abs_dprob_vs_default": dpd, "tolerance": 1e-5, "within_tolerance": bool(dpd <= 1e-5),
                "argmax_agreement": float((out_d["predictions"] == out["predictions"][:n]).mean()),
                "kernels": dkern,
                "trunk": ({"achieved": round(conv_d["tflops"], 2), "frac": round(conv_d["tflops"] / MFMA_PEAK_TFLOPS["float32"], 5),
                           "ms": round(conv_d["seconds"] * 1e3, 3)} if conv_d else None),
                "note": ("conv_algo='direct': the audit mode (float32 implicit GEMM for every block convolution, the reference's operation "
                         "order); `value` runs conv_algo='auto' -- Winograd F(2x2, 3x3), float32 in / float32 accumulate, on the 3x3 / "
                         "stride-1 layers, gated by the per-layer error-bound test (tests/test_engine.py::test_winograd_conv_matches_torch_cpu_fp32)")}
        if hw != 224:
            _, x224 = workload(224, n)
            run(x224)
            el224, o224, _ = timed(x224, k_extra, 1)
            assert o224["probabilities"].shape == (n, 9)
            extras["patch_224"] = {"value": round(n * k_extra / el224, 2), "unit": "patches/s",
                                   "ms_per_step": round(el224 / k_extra * 1e3, 3), "dtype": args.dtype,
                                   "workload": f"BASELINE configs[1]: {n} synthetic 224x224x3 patches, same call"}
            if args.dtype == "float32":
                run(x224, "float32", "direct")
                el224d, o224d, _ = timed(x224, k_extra, 1, "float32", "direct")
                engine.conv_algo = "auto"
                extras["patch_224"]["cnn_direct"] = {
                    "value": round(n * k_extra / el224d, 2), "unit": "patches/s", "ms_per_step": round(el224d / k_extra * 1e3, 3),
                    "max_abs_dprob_vs_default": float(np.abs(o224d["probabilities"].astype(np.float64)
                                                             - o224["probabilities"].astype(np.float64)).max())}
            del x224
        if not os.environ.get("TIA_BENCH_NO_CLASSIC"):
            # the HBM-bound kernels SURVEY 8(d) lists beside the headline (Reinhard, Otsu / morphological maskers, luminosity mask,
            # augmentation, Lab conversion): stage time, roofline fraction, counter traffic, one-line CPU baseline each
            import gc

            import bench_classic

            gc.collect()
            torch.cuda.empty_cache()
            t0 = time.perf_counter()
            extras["classic"] = bench_classic.classic(reps=10, cpu=not args.no_cpu_baseline)
            extras["classic"]["bench_wall_s"] = round(time.perf_counter() - t0, 1)
        if not os.environ.get("TIA_BENCH_NO_CONFIGS"):
            extras["configs"] = config_extras(args)
        line["extras"] = extras
    if not args.no_cpu_baseline and world_size == 1:
        cpu_model, _ = get_pretrained_model("resnet18-kather100k")
        line["cpu_baseline"] = cpu_baseline(target, host, cpu_model, args.cpu_sample)
    return line


def main() -> None:
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.config == "patch":
        line = bench_patch(args)
    else:
        import bench_configs

        line = getattr(bench_configs, f"bench_{args.config}")(args)
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
    try:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
