"""world_size-2 gloo tests (CPU) for the patch-sharded multi-process path."""

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


@pytest.mark.parametrize("n", [7, 8])
def test_all_gather_and_engine_world2(tmp_path, n):
    port = 29600 + (os.getpid() % 200) + n
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert p0.shape == (5, 9) and np.array_equal(p0, p1)
    from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor
    from tiatoolbox_amd.utils import synth

    single = PatchPredictor("resnet18-kather100k", batch_size=5).run(
        synth.g_he(5, 64, 64, seed=4), patch_mode=True, return_probabilities=True, patch_input_shape=(64, 64))
    np.testing.assert_allclose(p0, single["probabilities"], atol=1e-6)
