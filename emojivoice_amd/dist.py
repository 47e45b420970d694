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
from typing import Dict, Optional, Tuple

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


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_bounds(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous rows [lo, hi) of the global batch owned by ``rank`` (remainder spread over the first ranks)."""
    q, r = divmod(n_rows, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_padded_length(local_max_len: int, device=None) -> int:
    """Global Tp = max over ranks of the per-rank max mel length, rounded up to a multiple of 4
    (fix_len_compatibility, utils/model.py:14-20).  An 8-byte control-plane all-reduce, not a data-path collective."""
    t = torch.tensor([int(local_max_len)], dtype=torch.int64, device=device)
    if _world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    v = int(t.item())
    return (v + 3) // 4 * 4


def all_gather_rows(block: torch.Tensor, n_global: Optional[int] = None) -> torch.Tensor:
    """ONE all-gather that collates per-rank row blocks (rows_r, ...) sharded by ``shard_bounds`` into the global
    (n_global, ...) tensor in global row order, on every rank.  ``all_gather_into_tensor`` needs the same shape on every
    rank, so when ``n_global`` is not a multiple of the world size the shorter shards are padded to ceil(n / world) rows and
    the padding is dropped after the gather.  ``n_global=None`` asserts equal shards (no size exchange is made)."""
    world = _world()
    if world == 1:
        return block
    rank = dist.get_rank()
    block = block.contiguous()
    rows = block.shape[0]
    if n_global is None:
        n_global = rows * world
    lo, hi = shard_bounds(n_global, rank, world)
    if hi - lo != rows:
        raise ValueError(f"rank {rank} holds {rows} rows but shard_bounds({n_global}, {rank}, {world}) = [{lo}, {hi})")
    per = -(-n_global // world)
    if rows < per:
        pad = torch.zeros((per - rows,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        block = torch.cat([block, pad], dim=0)
    out = torch.empty((world * per,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block)
    if n_global == world * per:
        return out
    keep = []
    for r in range(world):
        l, h = shard_bounds(n_global, r, world)
        keep.append(out[r * per: r * per + (h - l)])
    return torch.cat(keep, dim=0)


def all_gather_waveforms(wav: torch.Tensor, n_global: Optional[int] = None) -> torch.Tensor:
    """The path's single exchange step: (B_local, 1, L) per-rank waveform blocks -> (B_global, 1, L) on every rank."""
    return all_gather_rows(wav, n_global)


@torch.inference_mode()
def synthesise_sharded(model, vocoder, x, x_lengths, n_timesteps, temperature=1.0, spks=None, length_scale=1.0, *,
                       z: Optional[torch.Tensor] = None, denoiser=None, denoiser_strength: float = 0.00025) -> Dict[str, torch.Tensor]:
    """Data-parallel ``MatchaTTS.synthesise`` + ``Generator.forward`` (+ clamp / denoiser of ``to_waveform``) over the ranks
    of the default process group.  Every rank is called with the SAME global batch (token ids are a few KB): it keeps rows
    ``shard_bounds(B, rank, world)``, runs the text encoder and durations for them, agrees on the GLOBAL padded length with
    the other ranks (8-byte MAX all-reduce), decodes and vocodes its shard at that length, and takes part in the one
    all-gather of the waveform blocks.  Padding to the global ``Tp`` and slicing this rank's rows out of ONE global noise
    draw (``z`` given, or the model's host draw for the whole batch, flow_matching.py:51) make the result equal to the
    single-process batch.  ``mel_lengths`` ride along as one extra column of the gathered block (exact in
    fp32 below 2^24 frames), so there is no second data collective.

    Returns ``wav`` (B, 1, 256 Tp) for the global batch, ``mel_lengths`` (B,) int64, ``Tp``, and this rank's ``mel`` shard."""
    world = _world()
    rank = dist.get_rank() if world > 1 else 0
    B = x.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    if hi == lo:
        raise ValueError(f"global batch of {B} utterances leaves rank {rank} of {world} without work")
    dev = model.device
    xs, xls = x[lo:hi], x_lengths[lo:hi]
    sp = spks[lo:hi] if spks is not None else None
    spk, mu_x, w_ceil, x_mask, xls_d, y_lengths, y_max_local = model._durations(xs, xls, sp, length_scale)
    if model.encoder_stage != "host":
        model.engine.text_encoder_status()
    Tp = global_padded_length(y_max_local, device=dev if (world > 1 and dist.get_backend() == "nccl") else None)
    if z is None:
        z = model.draw_noise(B, Tp)                       # ONE draw for the global batch, as the single-process run makes it
    if z.shape != (B, model.n_feats, Tp):
        raise ValueError(f"z must be the global draw of shape {(B, model.n_feats, Tp)}, got {tuple(z.shape)}")
    _, dec, mel, _ = model._decode_aligned(spk, mu_x, w_ceil, x_mask, xls_d, y_lengths, Tp, n_timesteps, temperature, z=z[lo:hi])
    wav = vocoder(mel).clamp(-1, 1)
    if denoiser is not None:
        wav = denoiser(wav.squeeze(1), strength=denoiser_strength).reshape(wav.shape)
    block = torch.cat([wav.reshape(hi - lo, -1), y_lengths.to(wav.dtype).reshape(-1, 1)], dim=1)
    full = all_gather_rows(block, B)
    return {"wav": full[:, :-1].unsqueeze(1), "mel_lengths": full[:, -1].round().long(), "Tp": Tp, "mel": mel,
            "rows": (lo, hi), "ranks": world}


def barrier():
    if _world() > 1:
        dist.barrier()
