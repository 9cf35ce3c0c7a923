"""gloo tests (CPU, world sizes 2 / 4 / 8) for the multi-process paths: uneven shards, ranks whose shard is empty (more
ranks than patches / patch rows / tiles), ragged instance tables, tile sharding, rank-local canvas bands."""

from __future__ import annotations

import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tiatoolbox_amd import distributed as tdist


def test_shard_bounds_cover_in_order():
    for n in (0, 1, 5, 8, 9, 4096, 4097):
        for p in (1, 2, 3, 8):
            spans = [tdist.shard_bounds(n, r, p) for r in range(p)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= -(-n // p) for lo, hi in spans)


def _worker(rank: int, world: int, port: int, n: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    lo, hi = tdist.shard_bounds(n, rank, world)
    full = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
    got = tdist.all_gather_rows(full[lo:hi].clone(), n)
    assert torch.equal(got, full)
    # engine: sharded run returns the same full result on every rank
    from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor
    from tiatoolbox_amd.utils import synth

    patches = synth.g_he(5, 64, 64, seed=4)
    eng = PatchPredictor("resnet18-kather100k", batch_size=2)
    res = eng.run(patches, patch_mode=True, return_probabilities=True, patch_input_shape=(64, 64))
    np.save(os.path.join(out_dir, f"p{rank}.npy"), res["probabilities"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize(("world", "n"), [(2, 7), (2, 8), (4, 7), (8, 3)])
def test_all_gather_and_engine(tmp_path, world, n):
    """``PatchPredictor`` over 5 patches on 2 / 4 / 8 ranks: shards of 3+2, 2+2+1+0 and 1+1+1+1+1+0+0+0 patches -- uneven
    shards and ranks with an EMPTY shard (they probe one patch for the row shape and contribute zero rows)."""
    port = 29600 + (os.getpid() % 200) + n + 16 * world
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    p0 = np.load(tmp_path / "p0.npy")
    assert p0.shape == (5, 9)
    for r in range(1, world):
        assert np.array_equal(p0, np.load(tmp_path / f"p{r}.npy"))
    from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor
    from tiatoolbox_amd.utils import synth

    single = PatchPredictor("resnet18-kather100k", batch_size=5).run(
        synth.g_he(5, 64, 64, seed=4), patch_mode=True, return_probabilities=True, patch_input_shape=(64, 64))
    np.testing.assert_allclose(p0, single["probabilities"], atol=1e-6)


# ------------------------------------------------------------------ instance tables across ranks
class _CpuInstanceNet(torch.nn.Module):
    """Deterministic two-head 'network' + oracle post-processing: lets the patch-sharded instance engine run on CPU
    (the real post-processing needs the GPU) so that the ragged all-gather of instance tables can be checked."""

    tasks = ("nuclei_segmentation",)

    def __init__(self) -> None:
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))
        self.preproc_func = lambda img: img
        self.postproc_func = self.postproc

    @staticmethod
    def infer_batch(model, batch_data, *, device):  # noqa: ARG004
        x = torch.as_tensor(batch_data).float()
        dark = (1.0 - x.mean(-1) / 255.0)[:, 8:56, 8:56]
        ramp = torch.linspace(-1, 1, 48)
        hv = torch.stack([ramp[None, None, :] * dark, ramp[None, :, None] * dark], dim=-1)
        return dark[..., None].contiguous(), hv.contiguous(), (1.0 + (dark > 0.8).float())[..., None].contiguous()

    def postproc(self, raw_maps, offset=(0, 0)):
        from oracle import hovernet as oh
        from tiatoolbox_amd.models.architecture.hovernet import HoVerNet

        npm, hv, tp = (np.asarray(m) for m in raw_maps)
        inst = oh.proc_np_hv(npm, hv)
        info = oh.get_instance_info(inst, np.around(tp).astype("uint8")[..., 0], offset)
        return (HoVerNet._pack(self, inst, info),)  # noqa: SLF001


def _instance_patches() -> np.ndarray:
    rng = np.random.default_rng(17)
    patches = np.full((5, 64, 64, 3), 235, np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    for k in range(5):
        for _ in range(k):  # patch 0 stays empty: an empty table must survive the gather
            cy, cx, r = rng.integers(14, 50), rng.integers(14, 50), rng.integers(4, 8)
            patches[k][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 20
    return patches


def _run_instance_engine():
    from tiatoolbox_amd.models.engine.io_config import IOInstanceSegmentorConfig
    from tiatoolbox_amd.models.engine.multi_task_segmentor import MultiTaskSegmentor

    res = {"units": "baseline", "resolution": 1.0}
    cfg = IOInstanceSegmentorConfig(input_resolutions=[res], output_resolutions=[res, res, res], patch_input_shape=[64, 64],
                                    patch_output_shape=[48, 48], stride_shape=[48, 48], margin=8, tile_shape=[96, 96])
    eng = MultiTaskSegmentor(_CpuInstanceNet(), batch_size=2, device="cpu")
    return eng.run(_instance_patches(), patch_mode=True, ioconfig=cfg, return_probabilities=True)


def _inst_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import pickle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    out = _run_instance_engine()
    with open(os.path.join(out_dir, f"inst{rank}.pkl"), "wb") as fh:
        pickle.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()


def _same_tables(a: dict, b: dict) -> None:
    assert set(a) == set(b)
    assert np.array_equal(a["predictions"], b["predictions"])
    for k in range(len(a["box"])):
        assert len(a["box"][k]) == len(b["box"][k])
        if len(a["box"][k]) == 0:
            continue
        assert np.array_equal(np.asarray(a["box"][k], dtype=np.int64), np.asarray(b["box"][k], dtype=np.int64))
        np.testing.assert_array_equal(np.asarray(a["centroid"][k], dtype=np.float64), np.asarray(b["centroid"][k], dtype=np.float64))
        assert [int(t) for t in a["type"][k]] == [int(t) for t in b["type"][k]]
        assert [float(p) for p in a["prob"][k]] == [float(p) for p in b["prob"][k]]
        for ca, cb in zip(a["contours"][k], b["contours"][k]):
            assert np.array_equal(ca, cb) and ca.dtype == np.int32
    for ha, hb in zip(a["probabilities"], b["probabilities"]):
        assert np.array_equal(ha, hb)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_instance_tables_gathered_across_ranks(tmp_path, world):
    """MultiTaskSegmentor patch mode on 2 / 4 / 8 ranks (gloo): every rank post-processes its own shard; label maps and the
    ragged instance tables (incl. an empty one and polygons of different lengths) are all-gathered in input order.  With 8
    ranks and 5 patches three ranks hold no patch at all and still take part in every collective."""
    import pickle

    port = 29900 + (os.getpid() % 150) + 7 * world
    mp.spawn(_inst_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    single = _run_instance_engine()
    assert sum(len(b) for b in single["box"]) >= 5 and len(single["box"][0]) == 0
    for rank in range(world):
        with open(tmp_path / f"inst{rank}.pkl", "rb") as fh:
            _same_tables(pickle.load(fh), single)  # noqa: S301


def _tile_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import pickle

    import test_tile_mode as ttm

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    gold = np.load(ttm.GOLD / "tile_golden.npz")
    eng, heads, _, wsi_shape = ttm._engine(gold, "a", ttm._OracleHoVerNet())  # noqa: SLF001
    eng.distributed = True
    out = eng._process_tile_mode([torch.from_numpy(h) for h in heads], wsi_shape, None, return_predictions=(True,))  # noqa: SLF001
    with open(os.path.join(out_dir, f"tile{rank}.pkl"), "wb") as fh:
        pickle.dump(out[0], fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tile_mode_sharded_across_ranks_matches_reference(tmp_path, world):
    """WSI tile mode on 2 / 4 ranks (gloo): the 25 tiles are post-processed round-robin inside every shape group (uneven
    shares), tables and tile label maps are gathered, and every rank ends with the table the REAL reference produces
    (tile_golden 'a')."""
    import pickle

    import test_tile_mode as ttm

    port = 30100 + (os.getpid() % 150) + 11 * world
    mp.spawn(_tile_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    gold = np.load(ttm.GOLD / "tile_golden.npz")
    for rank in range(world):
        with open(tmp_path / f"tile{rank}.pkl", "rb") as fh:
            ttm._check_table(pickle.load(fh), gold, "a")  # noqa: S301, SLF001


# ------------------------------------------------------------------ semantic WSI mode: rank-local bands
def _band_truth(h: int, w: int, c: int | None = None) -> torch.Tensor:
    y = torch.arange(h).view(h, 1)
    x = torch.arange(w).view(1, w)
    base = ((y * 31 + x * 7) % 251).to(torch.uint8)
    if c is None:
        return base
    return (base.float()[..., None] + torch.arange(c).float()) / 7.0


def _band_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    from tiatoolbox_amd.models.engine.semantic_segmentor import band_plan, exchange_bands

    for h, w, stride, oh in ((1000, 37, 90, 110), (333, 20, 450, 512), (95, 8, 10, 12)):
        row_ys = np.arange(0, int(np.ceil(h / stride) * stride), stride)
        plan = band_plan(row_ys, oh, h, rank, world)
        lo, hi = plan["own"]
        # own rows plus one leading row; a rank without rows of its own infers nothing at all
        assert plan["rows"] == (list(range(max(lo - 1, 0), hi)) if lo < hi else [])
        truth, truth_p = _band_truth(h, w), _band_truth(h, w, 3)
        y_lo, y_hi = plan["y_lo"], plan["y_hi"]
        got = exchange_bands(truth[y_lo:y_hi].clone(), plan, h)
        got_p = exchange_bands(truth_p[y_lo:y_hi].clone(), plan, h)
        assert torch.equal(got, truth) and torch.equal(got_p, truth_p), (h, rank)
        # the chunked form for HOST bands of slides that do not fit the device (CanvasBand streamed mode): same result
        from tiatoolbox_amd.models.engine.semantic_segmentor import exchange_bands_streamed

        for rows in (7, 4096):
            got_s = exchange_bands_streamed(truth_p[y_lo:y_hi].clone(), plan, h, torch.device("cpu"), rows=rows)
            assert torch.equal(got_s, truth_p), (h, rank, rows)
    torch.save(torch.tensor(plan["bands"]), os.path.join(out_dir, f"bands{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _band_mode_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    from types import SimpleNamespace

    from tiatoolbox_amd.models.engine.semantic_segmentor import (CanvasBand, _band_device_rows, band_plan, exchange_bands,
                                                                 exchange_bands_streamed, exchange_footprint)

    h, w, stride, oh = 1000, 37, 90, 110
    plan = band_plan(np.arange(0, int(np.ceil(h / stride) * stride), stride), oh, h, rank, world)
    maps = {"pred": ((), torch.uint8), "probs": ((3,), torch.float32)}
    assert exchange_footprint(plan, h, w, maps) == w * 3 * 4 * ((world + 1) * max(b[1] - b[0] for b in plan["bands"]) + h)
    # only rank 1 "must stream" (its engine forces a ring of 3 chunks; the others would stay resident): every rank streams, K = 3
    eng = SimpleNamespace(device_band_rows=3 if rank == 1 else None, memory_threshold=80)
    k = _band_device_rows(eng, CanvasBand.bytes_needed(plan["y_hi"] - plan["y_lo"], w, maps), torch.device("cpu"), world=world,
                          exchange_bytes=exchange_footprint(plan, h, w, maps))
    assert k == 3, (rank, k)
    # ... and nobody streams when nobody has to
    k0 = _band_device_rows(SimpleNamespace(device_band_rows=None, memory_threshold=80), 1 << 20, torch.device("cpu"), world=world)
    assert k0 is None
    # the agreed mode drives the SAME collective sequence on every rank
    truth = _band_truth(h, w, 3)
    mine = truth[plan["y_lo"]:plan["y_hi"]].clone()
    got = exchange_bands_streamed(mine, plan, h, torch.device("cpu"), rows=64) if k else exchange_bands(mine, plan, h)
    assert torch.equal(got, truth)
    torch.save(torch.tensor([k]), os.path.join(out_dir, f"k{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_band_mode_is_agreed_across_ranks(tmp_path):
    """resident vs streamed canvas bands used to be decided per rank (free memory, own band height) -- near the
    threshold ranks would issue different collective sequences in the band exchange.  Now one MAX all-reduce makes the choice
    common: with ONE of three ranks forced to stream, all three stream with its ring size, and the exchange completes."""
    world = 3
    port = 29850 + (os.getpid() % 100)
    mp.spawn(_band_mode_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert [int(torch.load(tmp_path / f"k{r}.pt")[0]) for r in range(world)] == [3, 3, 3]


@pytest.mark.parametrize("world_size", [2, 4, 8])
def test_semantic_band_exchange(tmp_path, world_size):
    """Collective logic of sharded semantic WSI inference on CPU tensors (gloo, 2 / 4 / 8 ranks): every rank owns a
    contiguous band of canvas rows, the bands tile the slide, and one padded all-gather rebuilds predictions and
    probabilities on every rank (rank-local bands instead of an all-reduce of a full-size map).  The 333-row
    slide has a single patch row: with more ranks than rows the surplus ranks own an empty band, infer nothing and still take
    part in the exchange."""
    from tiatoolbox_amd.models.engine.semantic_segmentor import band_plan

    for world in (1, 2, 3, 8):  # layout properties for any world size (no process group needed)
        for h, stride in ((20000, 450), (1000, 90), (95, 10), (40, 450)):
            row_ys = np.arange(0, int(np.ceil(h / stride) * stride), stride)
            plans = [band_plan(row_ys, stride + 62, h, r, world) for r in range(world)]
            bands = plans[0]["bands"]
            assert all(p["bands"] == bands for p in plans)
            covered = [b for b in bands if b[1] > b[0]]
            assert covered[0][0] == 0 and covered[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            assert sorted(r for p in plans for r in range(*p["own"])) == list(range(len(row_ys)))
            for p in plans:  # surplus ranks (more ranks than patch rows) get no inference work
                assert (p["rows"] == []) == (p["own"][0] >= p["own"][1])
    port = 29850 + (os.getpid() % 100) + 13 * world_size
    mp.spawn(_band_worker, args=(world_size, port, str(tmp_path)), nprocs=world_size, join=True)
    for r in range(1, world_size):
        assert torch.equal(torch.load(tmp_path / "bands0.pt"), torch.load(tmp_path / f"bands{r}.pt"))


# ------------------------------------------------------------------ output directory of WSI runs across ranks
def _save_dir_worker(rank: int, world: int, port: int, root: str) -> None:
    from pathlib import Path

    from tiatoolbox_amd.models.engine.engine_abc import outputs_written, prepare_engines_save_dir

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    root_p = Path(root)
    # 1. fresh directory: only rank 0 creates it, nobody raises, every rank gets the same path
    d = prepare_engines_save_dir(root_p / "run", patch_mode=False, overwrite=False, distributed=True)
    assert d == root_p / "run" and d.is_dir()
    if rank == 0:
        (d / "slide.npz").write_bytes(b"x")
    outputs_written(True)
    assert (d / "slide.npz").exists()          # the returned paths exist on every rank when run() returns
    dist.barrier()
    # 2. it exists now: EVERY rank raises FileExistsError (nobody is left waiting in a collective)
    try:
        prepare_engines_save_dir(root_p / "run", patch_mode=False, overwrite=False, distributed=True)
    except FileExistsError:
        pass
    else:
        raise AssertionError(f"rank {rank}: no FileExistsError")
    assert (d / "slide.npz").exists()
    dist.barrier()
    # 3. overwrite=True: removed and re-created ONCE (by rank 0); the old file is gone, the directory is there on all ranks
    d2 = prepare_engines_save_dir(root_p / "run", patch_mode=False, overwrite=True, distributed=True)
    assert d2.is_dir() and not (d2 / "slide.npz").exists()
    dist.barrier()
    # 4. the WSI-mode contract without save_dir is unchanged, and patch mode needs no directory
    try:
        prepare_engines_save_dir(None, patch_mode=False, distributed=True)
    except OSError as exc:
        assert "no save directory" in str(exc)
    assert prepare_engines_save_dir(None, patch_mode=True, distributed=True) is None
    # 5. the refusal keeps the attributes of the original exception: errno and filename travel, the message is not doubled
    try:
        prepare_engines_save_dir(root_p / "run", patch_mode=False, overwrite=False, distributed=True)
    except FileExistsError as exc:
        import errno as _errno

        assert exc.errno == _errno.EEXIST and str(exc.filename) == str(root_p / "run"), (exc.errno, exc.filename)
        assert str(exc).count("Errno") == 1, str(exc)
    else:
        raise AssertionError(f"rank {rank}: no FileExistsError")
    # 6. the WRITE phase: rank 0 writes; a failing write (here: the target's directory does not exist) is raised by EVERY rank
    #    instead of leaving the others in a barrier
    from tiatoolbox_amd.models.engine.engine_abc import write_outputs

    write_outputs(True, lambda: np.savez(d2 / "slide.npz", a=np.arange(3)))
    assert (d2 / "slide.npz").exists()
    try:
        write_outputs(True, lambda: np.savez(root_p / "missing" / "slide.npz", a=np.arange(3)))
    except FileNotFoundError as exc:
        assert "missing" in str(exc.filename)
    else:
        raise AssertionError(f"rank {rank}: the failed write was not raised")
    (root_p / f"ok{rank}").write_text("1")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_wsi_save_dir_is_prepared_by_rank_zero_only(tmp_path, world):
    """Every rank calls ``run(..., patch_mode=False, save_dir=...)``; the reference's ``mkdir(parents=True)`` /
    ``rmtree`` rule (``engine_abc.py:1832-1885``) must be applied once, by rank 0, with its outcome raised everywhere."""
    port = 29900 + (os.getpid() % 150) + world
    mp.spawn(_save_dir_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


# ------------------------------------------------------------------ several tasks per patch across ranks (HoVerNet+-like)
class _CpuTwoTaskNet(_CpuInstanceNet):
    """Nuclei table + a 'layer' table with DIFFERENT columns (no centroid / prob), like ``HoVerNetPlus.postproc``."""

    tasks = ("nuclei_segmentation", "layer_segmentation")

    def postproc(self, raw_maps, offset=(0, 0)):
        nuclei = super().postproc(raw_maps, offset)[0]
        npm = np.asarray(raw_maps[0])[..., 0]
        layer = (npm > 0.5).astype(np.uint8) + (npm > 0.9).astype(np.uint8)
        classes = [int(c) for c in np.unique(layer) if c]
        table = {"box": np.array([[0, 0, 48, 48]] * len(classes)).reshape(-1, 4),
                 "contours": [np.argwhere(layer == c)[:5].astype(np.int32) for c in classes],
                 "type": np.array(classes, dtype=np.uint8)}
        return nuclei, {"task_type": self.tasks[1], "predictions": layer, "info_dict": table, "seg_type": "semantic"}


def _run_two_task_engine():
    from tiatoolbox_amd.models.engine.io_config import IOInstanceSegmentorConfig
    from tiatoolbox_amd.models.engine.multi_task_segmentor import MultiTaskSegmentor

    res = {"units": "baseline", "resolution": 1.0}
    cfg = IOInstanceSegmentorConfig(input_resolutions=[res], output_resolutions=[res, res, res], patch_input_shape=[64, 64],
                                    patch_output_shape=[48, 48], stride_shape=[48, 48], margin=8, tile_shape=[96, 96])
    eng = MultiTaskSegmentor(_CpuTwoTaskNet(), batch_size=2, device="cpu")
    return eng.run(_instance_patches(), patch_mode=True, ioconfig=cfg, return_probabilities=True)


def _two_task_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import pickle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    out = _run_two_task_engine()
    with open(os.path.join(out_dir, f"two{rank}.pkl"), "wb") as fh:
        pickle.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_multi_task_tables_gathered_across_ranks(tmp_path, world):
    """Patch-sharded run of a model with SEVERAL tasks (the reference's multi-head path, ``multi_task_segmentor.py:1556-1730``):
    one sub-dict per task, every task's label maps and columns in input order on every rank -- 2 ranks (3 + 2 patches) and
    8 ranks (three of them without a patch)."""
    import pickle

    port = 29700 + (os.getpid() % 150) + 5 * world
    mp.spawn(_two_task_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    single = _run_two_task_engine()
    assert set(single) == {"nuclei_segmentation", "layer_segmentation", "probabilities"}
    for rank in range(world):
        with open(tmp_path / f"two{rank}.pkl", "rb") as fh:
            got = pickle.load(fh)  # noqa: S301
        assert set(got) == set(single)
        for task in ("nuclei_segmentation", "layer_segmentation"):
            a, b = got[task], single[task]
            assert set(a) == set(b) and a["seg_type"] == b["seg_type"]
            assert np.array_equal(a["predictions"], b["predictions"])
            for key in set(a) - {"predictions", "seg_type"}:
                assert len(a[key]) == len(b[key]) == 5
                for ra, rb in zip(a[key], b[key]):
                    if key == "contours":
                        assert len(ra) == len(rb) and all(np.array_equal(x, y) for x, y in zip(ra, rb))
                    else:
                        assert np.array_equal(np.asarray(ra, dtype=object), np.asarray(rb, dtype=object))
        for ha, hb in zip(got["probabilities"], single["probabilities"]):
            assert np.array_equal(ha, hb)


# ------------------------------------------------------------------ bench_configs' N > 1 paths, dry (CPU tensors, gloo)
def _bench_dry_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import argparse
    import json
    import sys
    from pathlib import Path

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench_configs as bc

    dev = torch.device("cpu")
    args = argparse.Namespace(steps=2, warmup=1)
    calls = []
    # the timing harness every config shares: W + K steps, max over ranks, each rank's own time kept for per_rank
    elapsed = bc._timed(lambda: calls.append(1), args, world, dev)  # noqa: SLF001
    assert len(calls) == 3 and elapsed > 0
    blocks = {}
    # vahadane: no collective -- own times only
    blocks["vahadane"] = bc._per_rank(world, dev, None, what="none")  # noqa: SLF001
    # semantic: the band all-gather of a 95-row x 40-column toy slide with 10 patch rows (uneven bands at world 2 / 3)
    out_b = np.array([[x, y, x + 12, y + 12] for y in range(0, 100, 10) for x in (0, 10, 20, 30)])
    blocks["semantic"] = bc.semantic_per_rank(out_b, 12, 95, rank, world, dev)
    # hovernet: label maps + ragged instance tables of the CPU instance engine's own (already gathered) result
    res = _run_instance_engine()
    blocks["hovernet"] = bc.hovernet_per_rank(res, rank, world, dev)
    for name, b in blocks.items():
        assert len(b["own_ms_per_step"]) == world and all(v >= 0 for v in b["own_ms_per_step"]), name
        assert ("collective_ms" in b) == (name != "vahadane")
        json.dumps(b)  # what bench.py prints must serialise
    if rank == 0:
        with open(os.path.join(out_dir, "per_rank.json"), "w") as fh:
            json.dump(blocks, fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_configs_multi_rank_paths_dry(tmp_path, world):
    """``bench.py --gpus N --config semantic|hovernet|vahadane``: the code that only runs at N > 1 -- the shared timing harness
    (own time per rank, max over ranks) and each config's ``per_rank`` block (own step times + the config's collective timed
    alone) -- executed under gloo on CPU tensors at toy size, so that an argument or gather bug cannot first appear on the
    8-GPU node.  (The GPU work of the configs themselves cannot run here: there is no CPU fallback.)"""
    import json

    port = 30300 + (os.getpid() % 150) + 17 * world
    mp.spawn(_bench_dry_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    blocks = json.loads((tmp_path / "per_rank.json").read_text())
    assert set(blocks) == {"vahadane", "semantic", "hovernet"}
    assert blocks["semantic"]["collective_bytes_per_rank"] > 0 and blocks["hovernet"]["collective_bytes_per_rank"] > 0


# ------------------------------------------------------------------ tile mode with several tasks per tile across ranks
class _OracleTwoTaskTiles:
    """Tile post-processing of a two-task model on the CPU: the oracle's nuclei pass plus a semantic 'layer' task with a uint8
    label map and its own columns -- drives ``_gather_tile_results``'s several-tasks branch (label maps as a flat int32 gather)."""

    tasks = ("nuclei_segmentation", "layer_segmentation")

    def postproc(self, maps, offset=(0, 0)):  # noqa: ARG002
        from oracle import hovernet as oh
        from tiatoolbox_amd.models.architecture.hovernet import HoVerNet

        npm, hv, tp = (np.asarray(m) for m in maps)
        inst = oh.proc_np_hv(npm, hv)
        info = oh.get_instance_info(inst, np.around(tp).astype("uint8")[..., 0])
        nuclei = HoVerNet._pack(self, inst, info)  # noqa: SLF001
        layer = ((npm[..., 0] > 0.5).astype(np.uint8) + (npm[..., 0] > 0.9).astype(np.uint8))
        rows = {"box": [], "centroid": [], "contours": [], "prob": [], "type": []}
        for c in (int(v) for v in np.unique(layer) if v):  # one record per layer class present in the tile
            ys, xs = np.nonzero(layer == c)
            rows["box"].append(np.array([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1]))
            rows["centroid"].append(np.array([xs.mean(), ys.mean()]))
            rows["contours"].append(np.stack([xs[:4], ys[:4]], axis=1).astype(np.int32))
            rows["prob"].append(1.0)
            rows["type"].append(c)
        table = {k: (np.array(v) if k in ("box", "centroid") and v else v) for k, v in rows.items()}
        return nuclei, {"task_type": self.tasks[1], "predictions": layer, "info_dict": table, "seg_type": "semantic"}


def _two_task_tiles(distributed: bool):
    import test_tile_mode as ttm

    gold = np.load(ttm.GOLD / "tile_golden.npz")
    eng, heads, _, wsi_shape = ttm._engine(gold, "a", _OracleTwoTaskTiles())  # noqa: SLF001
    eng.distributed = distributed
    return eng._process_tile_mode([torch.from_numpy(h) for h in heads], wsi_shape, None, return_predictions=(True, True))  # noqa: SLF001


def _two_task_tile_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import pickle

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    tdist.init_from_env("gloo")
    out = _two_task_tiles(True)
    with open(os.path.join(out_dir, f"tt{rank}.pkl"), "wb") as fh:
        pickle.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_mode_with_two_tasks_sharded_across_ranks(tmp_path):
    """WSI tile mode of a model with two tasks per tile on 3 ranks (gloo): the per-task tables travel as small host records, the
    tile label maps (int32 instance map, uint8 layer map) as ONE flat int32 ragged gather; every rank must end with exactly the
    single-process result -- the reference-checked nuclei table (tile_golden 'a') and both slide-sized maps, dtypes included."""
    import pickle

    import test_tile_mode as ttm

    port = 30500 + (os.getpid() % 150)
    mp.spawn(_two_task_tile_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    single = _two_task_tiles(False)
    gold = np.load(ttm.GOLD / "tile_golden.npz")
    ttm._check_table(single[0], gold, "a")  # noqa: SLF001
    assert single[1]["predictions"].dtype == np.uint8 and single[1]["predictions"].any()
    for rank in range(3):
        with open(tmp_path / f"tt{rank}.pkl", "rb") as fh:
            got = pickle.load(fh)  # noqa: S301
        ttm._check_table(got[0], gold, "a")  # noqa: SLF001
        for a, b in zip(got, single):
            assert a["predictions"].dtype == b["predictions"].dtype and np.array_equal(a["predictions"], b["predictions"])
