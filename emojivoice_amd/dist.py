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


def agree_max_length(local_max_len: int, local_error: int = 0, device=None) -> Tuple[int, int]:
    """(global max mel length, unrounded; global error flag): ONE 16-byte MAX all-reduce of [max(y_lengths) of this rank's rows,
    error flag] — control plane, not a data-path collective.  The error flag makes a rank-local failure (an out-of-range token id in
    one shard) COLLECTIVE: every rank learns of it here and raises, instead of one rank raising while the others block in the next
    collective until the backend's timeout."""
    t = torch.tensor([int(local_max_len), int(local_error)], dtype=torch.int64, device=device)
    if _world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    v = t.tolist()
    return int(v[0]), int(v[1])


def global_padded_length(local_max_len: int, device=None) -> int:
    """Global Tp = max over ranks of the per-rank max mel length, rounded up to a multiple of 4
    (fix_len_compatibility, utils/model.py:14-20)."""
    return (agree_max_length(local_max_len, 0, device)[0] + 3) // 4 * 4


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


class ShardState:
    """What a rank knows about its rows after the text encoder and the duration rounding (no collective has run yet)."""

    def __init__(self, rank, world, B, lo, hi, durations=None, error: Optional[BaseException] = None):
        self.rank, self.world, self.B, self.lo, self.hi = rank, world, B, lo, hi
        self.durations, self.error = durations, error

    @property
    def y_max_local(self) -> int:
        return int(self.durations[6]) if self.durations is not None else 0


@torch.inference_mode()
def shard_durations(model, x, x_lengths, spks, length_scale, rank: int, world: int) -> ShardState:
    """Per-rank compute, first half: text encoder + durations (matcha_tts.py:116-125) for rows ``shard_bounds(B, rank, world)`` of
    the GLOBAL batch.  No collective.  A token id outside the embedding table — the reference's ``IndexError`` at the lookup — is
    recorded in the returned state instead of raised, so that ``synthesise_sharded`` can raise it on EVERY rank after the
    agreement step."""
    B = x.shape[0]
    if B < world:            # the same value on every rank: everyone raises, nobody is left inside a collective
        raise ValueError(f"global batch of {B} utterances cannot be sharded over {world} ranks (every rank needs at least one)")
    lo, hi = shard_bounds(B, rank, world)
    sp = spks[lo:hi] if spks is not None else None
    try:
        d = model._durations(x[lo:hi], x_lengths[lo:hi], sp, length_scale)
        if model.encoder_stage != "host":
            model.engine.text_encoder_status()
    except IndexError as e:
        return ShardState(rank, world, B, lo, hi, None, e)
    return ShardState(rank, world, B, lo, hi, d)


@torch.inference_mode()
def shard_decode(model, vocoder, st: ShardState, y_max_global: int, z: Optional[torch.Tensor], n_timesteps, temperature=1.0, *,
                 denoiser=None, denoiser_strength: float = 0.00025):
    """Per-rank compute, second half: alignment, CFM decode and vocoder of this rank's rows at the GLOBAL length — ``Tp =
    fix_len_compatibility(y_max_global)`` frames inside the decoder (GroupNorm statistics and attention see the padded frames,
    decoder.py:41-43), mel trimmed to ``y_max_global`` as ``MatchaTTS.synthesise`` trims it (matcha_tts.py:146-152), so that the
    vocoder sees exactly the single-process ``out["mel"]``.  ``z``: the GLOBAL draw (B, n_feats, Tp), of which the rank takes
    its rows.  No collective.  Returns (block (rows, 256 * y_max_global + 1) = waveform | mel length, mel shard)."""
    from .text_encoder import fix_len_compatibility

    Tp = fix_len_compatibility(y_max_global)
    if z is None or tuple(z.shape) != (st.B, model.n_feats, Tp):
        raise ValueError(f"z must be the global draw of shape {(st.B, model.n_feats, Tp)}, got {None if z is None else tuple(z.shape)}")
    spk, mu_x, w_ceil, x_mask, xls_d, y_lengths, _ = st.durations
    _, dec, mel, _ = model._decode_aligned(spk, mu_x, w_ceil, x_mask, xls_d, y_lengths, y_max_global, n_timesteps, temperature, z=z[st.lo:st.hi])
    wav = vocoder(mel).clamp(-1, 1)
    if denoiser is not None:
        wav = denoiser(wav.squeeze(1), strength=denoiser_strength).reshape(wav.shape)
    block = torch.cat([wav.reshape(st.hi - st.lo, -1), y_lengths.to(wav.dtype).reshape(-1, 1)], dim=1)
    return block, mel


def collate_blocks(full: torch.Tensor, Tp: int, y_max: int, lo: int, hi: int, world: int, mel) -> Dict[str, torch.Tensor]:
    """The gathered (B, 256 * y_max + 1) block -> the result dict of ``synthesise_sharded``."""
    return {"wav": full[:, :-1].unsqueeze(1), "mel_lengths": full[:, -1].round().long(), "Tp": Tp, "y_max": y_max, "mel": mel,
            "rows": (lo, hi), "ranks": world}


@torch.inference_mode()
def synthesise_sharded(model, vocoder, x, x_lengths, n_timesteps, temperature=1.0, spks=None, length_scale=1.0, *,
                       z: Optional[torch.Tensor] = None, denoiser=None, denoiser_strength: float = 0.00025) -> Dict[str, torch.Tensor]:
    """Data-parallel ``MatchaTTS.synthesise`` + ``Generator.forward`` (+ clamp / denoiser of ``to_waveform``) over the ranks
    of the default process group.  Every rank is called with the SAME global batch (token ids are a few KB): it keeps rows
    ``shard_bounds(B, rank, world)``, runs the text encoder and durations for them (``shard_durations``), agrees on the GLOBAL
    maximum length with the other ranks (``agree_max_length``: one 16-byte MAX all-reduce that also carries an error flag),
    decodes and vocodes its shard at that length (``shard_decode``), and takes part in the one all-gather of the waveform
    blocks.  Decoding at the global ``Tp`` and slicing this rank's rows out of ONE global noise draw (``z`` given, or the model's
    host draw for the whole batch, flow_matching.py:51) make the result equal to the single-process batch: ``wav`` is
    ``vocoder(MatchaTTS.synthesise(global batch)["mel"])``.  ``mel_lengths`` ride along as one extra column of the
    gathered block (exact in fp32 below 2^24 frames), so there is no second data collective.
    Failures are collective: a batch smaller than the world size raises on every rank before any collective; an out-of-range
    token id in ONE shard raises ``IndexError`` on EVERY rank after the agreement step; a failure of the decode / vocoder stage on one
    rank (``EvLibraryError``, ``ValueError`` ...) is agreed on before the all-gather — that rank raises its exception, the others a
    ``RuntimeError`` — so no rank is ever left waiting inside a collective.

    Returns ``wav`` (B, 1, 256 * y_max) for the global batch, ``mel_lengths`` (B,) int64, ``Tp``, ``y_max`` and this rank's ``mel`` shard."""
    from .text_encoder import fix_len_compatibility

    world = _world()
    rank = dist.get_rank() if world > 1 else 0
    st = shard_durations(model, x, x_lengths, spks, length_scale, rank, world)
    dev = model.device
    y_max, err = agree_max_length(st.y_max_local, 1 if st.error is not None else 0,
                                  device=dev if (world > 1 and dist.get_backend() == "nccl") else None)
    if err:
        raise IndexError(str(st.error) if st.error is not None else "index out of range in self (a token id outside the embedding table on another rank)")
    Tp = fix_len_compatibility(y_max)
    if z is None:
        z = model.draw_noise(st.B, Tp)                    # ONE draw for the global batch, as the single-process run makes it
    # The decode stage can fail on ONE rank only (the library refusing a shape: "exceeds the 4 GiB limit: split the batch", out of memory,
    # a captured handle bound to another shape).  Left to propagate, that rank would raise while its peers block in the all-gather until
    # the backend's timeout: so the failure is caught, agreed on with one more 8-byte MAX all-reduce (control plane), and raised on every
    # rank — the failing rank re-raises its own exception, the others a RuntimeError that says which step failed elsewhere.
    local_exc: Optional[BaseException] = None
    block = mel = None
    try:
        block, mel = shard_decode(model, vocoder, st, y_max, z, n_timesteps, temperature, denoiser=denoiser, denoiser_strength=denoiser_strength)
    except Exception as e:  # noqa: BLE001 - every failure of the stage must reach the agreement below
        local_exc = e
    if world > 1:
        flag = torch.tensor([1 if local_exc is not None else 0], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else None)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = bool(flag.item())
    else:
        failed = local_exc is not None
    if local_exc is not None:
        raise local_exc
    if failed:
        raise RuntimeError("synthesise_sharded: the decode / vocoder stage failed on another rank (its exception is raised there); no waveform was gathered")
    full = all_gather_rows(block, st.B)
    return collate_blocks(full, Tp, y_max, st.lo, st.hi, world, mel)


def barrier():
    if _world() > 1:
        dist.barrier()
