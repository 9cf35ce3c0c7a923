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
