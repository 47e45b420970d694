"""The N > 1 data-parallel path on CPU (gloo, world size 2): row sharding incl. uneven shards, the single padded
all-gather, and ``dist.synthesise_sharded`` end to end.

The HIP stages cannot run here, so ``synthesise_sharded`` is driven with stand-ins for the model / vocoder that compute
with the CPU oracle (test infrastructure).  What is under test is the sharding logic itself: per-rank rows, the GLOBAL
padded length, per-rank slices of ONE global noise draw, the gather order and the trimmed padding — the sharded result
must equal the oracle's single-process run of the whole batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(worker, world=2, timeout=300):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = []
    for _ in range(world):
        item = q.get(timeout=timeout)
        if isinstance(item, str):                              # a worker failed: stop the others and show its traceback
            for p in ps:
                p.terminate()
            raise AssertionError(item)
        res.append(item)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def _guard(fn):
    """Run a worker body; a failure is reported through the queue instead of leaving the parent waiting for its timeout."""
    import functools
    import traceback

    @functools.wraps(fn)
    def run(rank, world, port, q):
        try:
            fn(rank, world, port, q)
        except BaseException:  # noqa: BLE001
            q.put(f"rank {rank} failed:\n{traceback.format_exc()}")
            raise
    return run


def _init(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from emojivoice_amd import dist as D

    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    return D


def _gather_body(rank, world, port, q):
    D = _init(rank, world, port)
    out = []
    for B in (6, 7, 3):                                        # even shards, uneven shards (4 + 3), fewer rows than 2 per rank
        lengths = torch.tensor([40, 13, 77, 5, 61, 130, 9][:B])
        lo, hi = D.shard_bounds(B, rank, world)
        tp = D.global_padded_length(int(lengths[lo:hi].max()))
        # stand-in for the vocoder output of this shard: value encodes (global row, sample index)
        wav = torch.stack([torch.arange(tp * 4, dtype=torch.float32) + 1000.0 * i for i in range(lo, hi)]).unsqueeze(1)
        full = D.all_gather_waveforms(wav, B)
        out.append((B, tp, tuple(full.shape), [float(v) for v in full[:, 0, 0]], float(full[-1, 0, -1])))
    try:                                                         # a shard that contradicts shard_bounds must be refused, not gathered
        D.all_gather_rows(torch.zeros(5, 2), 7)
        out.append("no error")
    except ValueError:
        out.append("refused")
    D.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def _worker_gather(rank, world, port, q):
    _guard(_gather_body)(rank, world, port, q)


def _worker_synth(rank, world, port, q):
    _guard(_synth_body)(rank, world, port, q)


def test_two_rank_shard_and_gather():
    for rank, out in _spawn(_worker_gather):
        for B, want_tp, (b, tp, shape, firsts, last) in zip((6, 7, 3), (132, 132, 80), out[:3]):
            assert b == B and tp == want_tp                    # global max rounded up to a multiple of 4, identical on both ranks
            assert shape == (B, 1, tp * 4)                     # rows in global order, padding rows dropped
            assert firsts == [1000.0 * i for i in range(B)]
            assert last == 1000.0 * (B - 1) + tp * 4 - 1
        assert out[3] == "refused"


class _OracleModel:
    """Stand-in for emojivoice_amd.MatchaTTS with the two internal stages synthesise_sharded drives, computed by the oracle."""

    def __init__(self, sd):
        from oracle import matcha_oracle as O

        self.O, self.sd = O, sd
        self.device = torch.device("cpu")
        self.n_feats = 80
        self.encoder_stage = "host"
        self.n_spks = sd["spk_emb.weight"].shape[0]

    def draw_noise(self, B, Tp):
        return torch.randn_like(torch.empty(B, Tp, self.n_feats).transpose(1, 2))

    def _durations(self, x, x_lengths, spks, length_scale):
        O = self.O
        spk = torch.nn.functional.embedding(spks.long(), self.sd["spk_emb.weight"])
        mu_x, logw, x_mask = O.text_encoder(self.sd, x, x_lengths, spk)
        w_ceil = torch.ceil(torch.exp(logw) * x_mask) * length_scale
        y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
        return spk, mu_x, w_ceil, x_mask, x_lengths, y_lengths, int(y_lengths.max())

    def _decode_aligned(self, spk, mu_x, w_ceil, x_mask, x_lengths, y_lengths, y_max_length, n_timesteps, temperature, z=None):
        O = self.O
        Tp = O.fix_len_compatibility(y_max_length)
        y_mask = O.sequence_mask(y_lengths, Tp).unsqueeze(1).to(x_mask.dtype)
        attn = O.generate_path(w_ceil.squeeze(1), (x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1)).unsqueeze(1)
        mu_y = torch.matmul(attn.squeeze(1).transpose(1, 2), mu_x.transpose(1, 2)).transpose(1, 2)
        dec = O.cfm_decode(self.sd, mu_y, y_mask, n_timesteps, temperature, spk, z=z)
        mel = O.denormalize(dec, self.sd["mel_mean"], self.sd["mel_std"])
        return mu_y, dec[:, :, :y_max_length], mel[:, :, :y_max_length], attn


def _synth_body(rank, world, port, q):
    torch.set_num_threads(2)
    D = _init(rank, world, port)
    from emojivoice_amd import weights as W
    from oracle import matcha_oracle as O

    sd, voc_sd = W.synthetic_matcha_state(178, 109), W.synthetic_hifigan_state()
    g = torch.Generator().manual_seed(99)
    B, Lx = 5, 14                                              # uneven: rank 0 holds 3 utterances, rank 1 holds 2
    ids = torch.randint(1, 178, (B, Lx), generator=g)
    xl = torch.tensor([14, 6, 9, 3, 12])
    spks = torch.tensor([107, 58, 0, 12, 17])
    model = _OracleModel(sd)
    vocoder = lambda mel: O.hifigan_forward(voc_sd, mel, W.HIFIGAN_V1)   # noqa: E731
    torch.manual_seed(4242)                                    # same seed on every rank -> the same GLOBAL draw
    out = D.synthesise_sharded(model, vocoder, ids, xl, 2, 0.667, spks, 1.0)
    # single-process reference run of the whole batch with the same seed (the oracle's own synthesise) and the vocoder on ITS mel
    torch.manual_seed(4242)
    ref = O.synthesise(sd, ids, xl, 2, 0.667, spks, 1.0)
    Tp = O.fix_len_compatibility(int(ref["mel_lengths"].max()))
    lo, hi = out["rows"]
    e_mel = float((out["mel"] - ref["mel"][lo:hi]).abs().max()) if out["mel"].shape == ref["mel"][lo:hi].shape else 1e9
    ref_wav = O.hifigan_forward(voc_sd, ref["mel"], W.HIFIGAN_V1).clamp(-1, 1)          # = to_waveform(synthesise(...)["mel"])
    e_wav = float((out["wav"] - ref_wav).abs().max()) if out["wav"].shape == ref_wav.shape else 1e9
    same_len = bool(torch.equal(out["mel_lengths"], ref["mel_lengths"]))
    # a token id outside the embedding table in ONE shard (rank 1's rows) must raise IndexError on EVERY rank, not hang the others
    bad = ids.clone()
    bad[4, 2] = 100000
    try:
        D.synthesise_sharded(model, vocoder, bad, xl, 2, 0.667, spks, 1.0)
        collective_error = "no error"
    except IndexError:
        collective_error = "IndexError"
    try:                                                       # fewer utterances than ranks: every rank refuses before any collective
        D.synthesise_sharded(model, vocoder, ids[:1], xl[:1], 2, 0.667, spks[:1], 1.0)
        small = "no error"
    except ValueError:
        small = "ValueError"
    # a failure of the decode / vocoder stage on ONE rank (here: rank 1's vocoder raises, as the library does for an over-size tensor):
    # rank 1 raises its own exception, rank 0 a RuntimeError — neither is left inside the all-gather
    def flaky_vocoder(mel):
        if rank == 1:
            raise ValueError("tensor exceeds the 4 GiB buffer-addressing limit: split the batch")
        return vocoder(mel)
    try:
        torch.manual_seed(4242)
        D.synthesise_sharded(model, flaky_vocoder, ids, xl, 2, 0.667, spks, 1.0)
        stage_error = "no error"
    except ValueError as e:
        stage_error = "ValueError" if "4 GiB" in str(e) else "other ValueError"
    except RuntimeError as e:
        stage_error = "RuntimeError" if "another rank" in str(e) else "other RuntimeError"
    # ... and the group is still usable afterwards
    torch.manual_seed(4242)
    again = D.synthesise_sharded(model, vocoder, ids, xl, 2, 0.667, spks, 1.0)
    usable = bool(torch.equal(again["wav"], out["wav"]))
    q.put((rank, out["Tp"], Tp, tuple(out["wav"].shape), int(ref["mel_lengths"].max()), e_mel, e_wav, same_len, out["ranks"], collective_error, small, stage_error, usable))
    D.barrier()
    dist.destroy_process_group()


def test_synthesise_sharded_equals_single_process():
    for rank, tp, want_tp, shape, y_max, e_mel, e_wav, same_len, ranks, collective_error, small, stage_error, usable in _spawn(_worker_synth, timeout=600):
        assert ranks == 2 and tp == want_tp
        assert shape == (5, 1, 256 * y_max)                    # the vocoder saw the mel trimmed to max(y_lengths), as synthesise returns it
        assert same_len                                        # mel_lengths ride along in the one gather, exact
        assert e_mel <= 1e-5 and e_wav <= 1e-5, (rank, e_mel, e_wav)
        assert collective_error == "IndexError", (rank, collective_error)   # raised on BOTH ranks (the bad id sits in rank 1's shard)
        assert small == "ValueError"
        assert stage_error == ("ValueError" if rank == 1 else "RuntimeError"), (rank, stage_error)   # the failing rank: its own exception; the other: told so
        assert usable
