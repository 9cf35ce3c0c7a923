"""One-process-per-GPU sharding of patch batches (RCCL over xGMI via ``torch.distributed``).

The path shards embarrassingly: every patch's normalisation / classification is
independent.  Rank ``r`` of ``P`` owns the contiguous slice
``[r*ceil(N/P), (r+1)*ceil(N/P))`` so that a gather in rank order restores input order.
The only collective is an ``all_gather`` of the small per-patch results (``[N/P, C]``
probabilities, ~18 KB per rank for BASELINE config 2): latency-bound, one call per run.
Backend ``"nccl"`` is RCCL on ROCm; the CPU tests use ``gloo``.
"""

from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE, ...)."""
    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world_size,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)
    return rank, world_size, local_rank


def agree_max(values: list[int], device: torch.device) -> list[int]:
    """Element-wise MAX of a few integers over the ranks (one small all-reduce; identity without a process group): how ranks
    agree on a decision that each of them would otherwise take from rank-local quantities (free memory, band height)."""
    if not is_distributed() or dist.get_world_size() == 1:
        return [int(v) for v in values]
    on = device if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.cpu().tolist()]


def shard_bounds(n: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous ownership: ``[lo, hi)`` of rank ``rank`` (possibly empty for trailing ranks)."""
    per = math.ceil(n / world_size) if n else 0
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Gather row-sharded results (sharded with :func:`shard_bounds`) back into input order.

    Pads every shard to ``ceil(N/P)`` rows so a single fixed-size ``all_gather_into_tensor``
    suffices, then trims.  Returns ``local`` unchanged when not distributed.
    """
    if not is_distributed():
        return local
    rank, world_size = world()
    per = math.ceil(n_total / world_size)
    pad = torch.zeros((per, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world_size * per, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    return out[:n_total]


def all_gather_ragged(local: torch.Tensor) -> tuple[torch.Tensor, list[int]]:
    """Concatenate per-rank tensors of different length along dim 0, in rank order.

    One fixed-size ``all_gather`` of the lengths, then one ``all_gather_into_tensor`` of the payload padded to the
    longest shard.  Returns ``(concatenated, lengths per rank)``; ``(local, [len])`` when not distributed.
    """
    if not is_distributed():
        return local, [int(local.shape[0])]
    _, world_size = world()
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    lens = torch.empty(world_size, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(lens, n_local)
    lens_h = [int(v) for v in lens.cpu().tolist()]
    per = max(max(lens_h), 1)
    pad = torch.zeros((per, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world_size * per, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous())
    parts = [out[r * per: r * per + lens_h[r]] for r in range(world_size)]
    return torch.cat(parts), lens_h


def all_gather_objects(obj) -> list:
    """One picklable host object per rank, returned in rank order on every rank (``[obj]`` when not distributed).

    For the small, irregular host-side records of multi-task post-processing (HoVerNet+: nuclei table + layer table per
    patch / tile, with different columns per task); on the ``nccl`` backend the pickled bytes travel through RCCL as byte
    tensors on the current device.  The single-task hot path uses the packed numeric gather below instead."""
    if not is_distributed():
        return [obj]
    _, world_size = world()
    out: list = [None] * world_size
    dist.all_gather_object(out, obj)
    return out


_TABLE_KEYS = ("box", "centroid", "contours", "prob", "type")


def gather_instance_tables(tables: list[dict], device: torch.device) -> list[dict]:
    """All-gather per-patch instance tables (``box`` / ``centroid`` / ``contours`` / ``prob`` / ``type`` columns)
    of patch-sharded post-processing, SURVEY section 8(e): counts first, then the flat payloads -- instance counts per
    patch, boxes, centroids, types, probabilities, polygon lengths and one packed vertex list.  Every rank gets the
    tables of all patches in input order (shards are contiguous and gathered in rank order)."""
    import numpy as np

    if not is_distributed():
        return tables
    n_inst = [len(t["box"]) for t in tables]
    k = sum(n_inst)

    def col(key, shape, dtype, conv=None):
        rows = [np.asarray(conv(t[key]) if conv else t[key]).reshape(-1, *shape) for t, c in zip(tables, n_inst) if c]
        return np.concatenate(rows).astype(dtype) if rows else np.zeros((0, *shape), dtype)

    none_to = lambda fill: (lambda c: [fill if v is None else v for v in c])  # noqa: E731
    polys = [p for t, c in zip(tables, n_inst) if c for p in t["contours"]]
    payload = {
        "count": np.asarray(n_inst, np.int64).reshape(-1),
        "box": col("box", (4,), np.int64),
        "centroid": col("centroid", (2,), np.float64),
        "type": col("type", (), np.int64, none_to(-1)),
        "prob": col("prob", (), np.float64, none_to(float("nan"))),
        "polylen": np.asarray([len(p) for p in polys], np.int64).reshape(-1),
        "poly": np.concatenate(polys).astype(np.int32) if polys else np.zeros((0, 2), np.int32),
    }
    assert payload["box"].shape[0] == k
    got = {name: all_gather_ragged(torch.from_numpy(np.ascontiguousarray(arr)).to(device))[0].cpu().numpy()
           for name, arr in payload.items()}
    out, i0, p0 = [], 0, 0
    for c in got["count"].tolist():
        if c == 0:
            empty = np.empty(shape=0)
            out.append({key: empty for key in _TABLE_KEYS})
            continue
        lens = got["polylen"][i0:i0 + c]
        contours = np.empty(c, dtype=object)
        for j, m in enumerate(lens.tolist()):
            contours[j] = got["poly"][p0:p0 + m]
            p0 += m
        types = np.empty(c, dtype=object)
        probs = np.empty(c, dtype=object)
        for j in range(c):
            tv, pv = int(got["type"][i0 + j]), float(got["prob"][i0 + j])
            types[j] = None if tv < 0 else tv
            probs[j] = None if pv != pv else pv
        out.append({"box": got["box"][i0:i0 + c], "centroid": got["centroid"][i0:i0 + c], "contours": contours,
                    "prob": probs, "type": types})
        i0 += c
    return out
