"""Data-parallel sharding of emoji-tagged utterance batches: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The path shards by utterance with no data-path collective; the one exchange step is a
single all-gather that collates the per-rank waveforms (north_star config 3).  The
reference has no counterpart (its inference is single-device).

Numerics note: results for an utterance depend on the padded length ``Tp`` of its batch
(GroupNorm statistics and attention see padded frames), so every rank pads to the
GLOBAL ``Tp``; a W-way run is then identical to the single-process batch.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's env; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_bounds(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous rows [lo, hi) of the global batch owned by ``rank`` (remainder spread over the first ranks)."""
    q, r = divmod(n_rows, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_padded_length(local_max_len: int, device=None) -> int:
    """Global Tp = max over ranks of the per-rank max mel length, rounded up to a multiple of 4."""
    t = torch.tensor([int(local_max_len)], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    v = int(t.item())
    return (v + 3) // 4 * 4


def all_gather_waveforms(wav: torch.Tensor) -> torch.Tensor:
    """One all-gather of the equally-shaped per-rank waveform blocks (B_local, 1, L) -> (W * B_local, 1, L)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return wav
    world = dist.get_world_size()
    wav = wav.contiguous()
    out = torch.empty((world * wav.shape[0],) + tuple(wav.shape[1:]), dtype=wav.dtype, device=wav.device)
    dist.all_gather_into_tensor(out, wav)
    return out


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
