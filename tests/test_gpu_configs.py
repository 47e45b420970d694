"""BASELINE.json configs at their real sizes on a MI355X, through the C ABI, against the CPU oracle:

  config 2  batch 64 x 516 frames, 10 Euler steps + HiFi-GAN — the tensors bench.py times, on the schedule it times
            (two-stream BatchPipeline) and back to back; rows vs the oracle and vs the oracle-checked B = 8 decode
  config 4  ODE-step sweep: n in {2, 4, 10, 20, 50} (+ 100: beyond the old 64-step plan) vs the oracle at the same n
  config 5  the feel_me.py streaming loop: B = 1, mixed lengths, all 11 emojis + an unmapped one, SPEAKING_RATE 0.8,
            10 steps, temperature 0.667, clamp + denoiser — each utterance vs the oracle
plus the engine-robustness cases of this round (two handles on two host threads, token-id validation, short denoiser input).
"""
import threading

import numpy as np
import pytest
import torch

import bench
from emojivoice_amd import weights as W
from oracle import matcha_oracle as O

pytestmark = pytest.mark.gpu
MEL_GATE = 1e-4
WAV_RMS_GATE = 1e-3
DEV = "cuda:0"


def _linf(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max())


def _rms(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(b).double()).pow(2).mean().sqrt())


@pytest.fixture(scope="module")
def sds():
    return W.synthetic_matcha_state(), W.synthetic_hifigan_state()


@pytest.fixture(scope="module")
def model(sds):
    from emojivoice_amd.matcha_tts import MatchaTTS

    return MatchaTTS(sds[0], device=DEV)


@pytest.fixture(scope="module")
def vocoder(sds):
    from emojivoice_amd.hifigan import AttrDict, Generator, v1

    g = Generator(AttrDict(v1)).to(DEV)
    g.load_state_dict(sds[1])
    return g


# ---------------------------------------------------------------------------------------------------------------------
# config 2 at its real size
# ---------------------------------------------------------------------------------------------------------------------
def test_config2_full_batch_vs_oracle(model, vocoder, sds):
    """B = 64 x T = 516 selects tile configurations a small batch never reaches (deep-grid 128x128 / 64x192 builds, the
    start stagger): decode + vocode exactly bench.py's tensors, both on the timed two-stream schedule and back to back, and
    compare rows with the CPU oracle.  Every length equals Tp, so rows do not interact in the reference either
    (GroupNorm and attention are per utterance) and the oracle can be run on a few rows."""
    from emojivoice_amd.pipeline import BatchPipeline

    sd, voc_sd = sds
    B, T, n_ode = 64, 516, 10
    mu, z, spk_ids, lengths = bench.make_inputs(B, T, 0, B, torch.device(DEV))
    spk = model._sd["spk_emb.weight"][spk_ids]
    z = z * 0.667
    mel = model.engine.cfm_decode(mu, lengths, spk, z, n_ode, model.mel_std, model.mel_mean)
    wav = vocoder(mel)
    pipe = BatchPipeline(model, vocoder)
    wav_p = [pipe.submit(mu, lengths, spk, z, n_ode) for _ in range(3)]     # three batches in flight: cfm(i+1) || hifigan(i)
    pipe.synchronize()
    for w in wav_p:
        assert torch.equal(w, wav)                                          # the timed schedule gives the same bits
    assert wav.shape == (B, 1, 256 * T) and bool(torch.isfinite(wav).all())
    rows = [0, 1, 62, 63]
    r = torch.tensor(rows)
    ref_mel, ref_wav = bench.oracle_rows(sd, voc_sd, mu.cpu()[r], z.cpu()[r], spk.cpu()[r], n_ode)
    assert _linf(mel[r.to(DEV)], ref_mel) <= MEL_GATE
    assert _rms(wav[r.to(DEV)], ref_wav) <= WAV_RMS_GATE / 10
    assert _linf(wav[r.to(DEV)], ref_wav) <= 1e-3
    assert float(ref_wav.pow(2).mean().sqrt()) > 0.05                       # the gate is not vacuous
    # rows 0..7 against the same rows decoded as their own B = 8 batch (the slice test_gpu_parity checks row-independence on)
    mel8 = model.engine.cfm_decode(mu[:8], lengths[:8], spk[:8], z[:8], n_ode, model.mel_std, model.mel_mean)
    wav8 = vocoder(mel8)
    assert _linf(mel[:8], mel8.cpu()) <= 1e-5
    assert _linf(wav[:8], wav8.cpu()) <= 5e-5


# ---------------------------------------------------------------------------------------------------------------------
# config 4: ODE-step sweep
# ---------------------------------------------------------------------------------------------------------------------
def test_config4_ode_sweep_vs_oracle(model, sds):
    sd = sds[0]
    g = torch.Generator().manual_seed(404)
    B, Tp = 2, 32
    mu = torch.randn(B, 80, Tp, generator=g)
    z = torch.randn(B, 80, Tp, generator=g)
    lengths = torch.tensor([32, 23])
    mask = O.sequence_mask(lengths, Tp).unsqueeze(1).float()
    spk = sd["spk_emb.weight"][torch.tensor([15, 54])]
    outs, refs = {}, {}
    for n in (2, 4, 10, 20, 50, 100):
        refs[n] = O.cfm_decode(sd, mu * mask, mask, n, 0.667, spk, z=z)
        outs[n], _ = model.decode((mu * mask).cuda(), lengths.cuda(), n, 0.667, spk.cuda(), z=z.cuda())
        assert _linf(outs[n], refs[n]) <= MEL_GATE / 2, n
    # the sweep's quality axis: mel-MSE against the 50-step output, GPU vs the same quantity on the CPU restatement
    for n in (2, 4, 10, 20):
        mse_gpu = float(((outs[n].cpu() - outs[50].cpu()) ** 2).mean())
        mse_cpu = float(((refs[n] - refs[50]) ** 2).mean())
        assert abs(mse_gpu - mse_cpu) <= 1e-6 + 1e-3 * mse_cpu, (n, mse_gpu, mse_cpu)
    assert float(((refs[2] - refs[50]) ** 2).mean()) > float(((refs[20] - refs[50]) ** 2).mean())   # more steps, closer to n = 50


# ---------------------------------------------------------------------------------------------------------------------
# config 5: the streaming feel_me.py loop
# ---------------------------------------------------------------------------------------------------------------------
RESPONSES = [
    "I love this so much \U0001F60D",
    "That makes me really angry \U0001F621 please stop doing it right now",
    "\U0001F60E cool",
    "Oh no that is so sad \U0001F62D I am sorry to hear about your day",
    "Sure (whatever you say) \U0001F644",
    "What a wonderful surprise \U0001F601 thank you",
    "Hello there \U0001F642 how are you doing today? I hope everything is going well with you and your family",
    "Ha ha ha that is hilarious \U0001F923",
    "Wow \U0001F62E",
    "Well that was close \U0001F605 we nearly missed the train this morning",
    "Hmm let me think about that for a moment \U0001F914 it is a difficult question",
    "Hello world \U0001F60A",                       # config 1's text: the emoji is not one of the 11 -> speaker 0
    "\U0001F642",                                   # only an emoji -> says 'nice' (feel_me.py:316-317)
    "first wins \U0001F62D then \U0001F60D",
]


def test_config5_streaming_loop_vs_oracle(model, vocoder, sds):
    from emojivoice_amd import emoji as E
    from emojivoice_amd import streaming as S
    from emojivoice_amd.denoiser import Denoiser
    from emojivoice_amd.emoji import EMOJI_MAPPING

    sd, voc_sd = sds
    den = Denoiser(vocoder, mode="zeros")
    tts = S.EmojiTTS(model, vocoder, den, text_to_ids=S.table_front_end)
    bias = O.denoiser_bias_spec(voc_sd, W.HIFIGAN_V1)
    seen, lens = set(), []
    for i, resp in enumerate(RESPONSES):
        torch.manual_seed(5000 + i)
        out = tts.respond(resp)
        text, spk = O.parse_emoji_response(resp, E.is_emoji, E.replace_emoji)
        assert (out["text"], out["spk"]) == (text, spk)
        seen.add(spk)
        ids = torch.tensor(S.T.intersperse(S.table_front_end(text.strip()), 0))[None]
        assert torch.equal(out["x"].cpu(), ids)
        torch.manual_seed(5000 + i)                 # seed parity: the product draws z exactly as the reference CPU run does
        ref = O.synthesise(sd, ids, torch.tensor([ids.shape[1]]), S.STEPS, S.TTS_TEMPERATURE, torch.tensor([spk]), S.SPEAKING_RATE)
        if not torch.equal(out["mel_lengths"].cpu(), ref["mel_lengths"]):          # integer output: exact; say which stage disagreed
            spk_e = torch.nn.functional.embedding(torch.tensor([spk]), sd["spk_emb.weight"])
            _, rlogw, rmask = O.text_encoder(sd, ids, torch.tensor([ids.shape[1]]), spk_e)
            _, dlogw = model.engine.text_encoder(ids.cuda(), torch.tensor([ids.shape[1]]).cuda(), spk_e.cuda())
            nflip = int((torch.ceil(torch.exp(rlogw)) != torch.ceil(torch.exp(dlogw.cpu()))).sum())
            raise AssertionError(f"mel_lengths {out['mel_lengths'].tolist()} vs {ref['mel_lengths'].tolist()} for {resp!r}: {nflip} ceil() flips, "
                                 f"max |dlogw| {float((rlogw - dlogw.cpu()).abs().max()):.2e}")
        assert tuple(out["mel"].shape) == tuple(ref["mel"].shape)
        assert _linf(out["mel"], ref["mel"]) <= MEL_GATE, resp
        ref_wav = O.to_waveform(voc_sd, W.HIFIGAN_V1, ref["mel"], bias)
        assert tuple(out["waveform"].shape) == tuple(ref_wav.shape) == (256 * ref["mel"].shape[-1],)
        assert _rms(out["waveform"], ref_wav) <= WAV_RMS_GATE / 10, resp
        assert _linf(out["waveform"], ref_wav) <= 1e-3, resp
        lens.append(int(ref["mel_lengths"][0]))
    assert seen == set(EMOJI_MAPPING.values()) | {0}            # all 11 emoji voices + the default speaker
    assert min(lens) < 40 and max(lens) > 300                   # mixed lengths


# ---------------------------------------------------------------------------------------------------------------------
# durations: the device text encoder must give the host stage's integer mel lengths
# ---------------------------------------------------------------------------------------------------------------------
def test_device_encoder_duration_flips(model, sds):
    """``mel_lengths`` are integers (``ceil(exp(logw))`` summed and truncated, matcha_tts.py:122-124): bit-exact against the CPU
    reference is the bar.  Over 200 random id sequences the device text encoder (``ev_text_encoder``) is compared with the CPU
    oracle's text encoder: per-token ``ceil()`` and, through the product's own ``_durations``, the integer ``mel_lengths`` at
    length scales 1.0 and 0.8.  A flip needs ``exp(logw)`` within ~1e-6 (relative) of an integer; the count is printed and must be 0."""
    g = torch.Generator().manual_seed(2024)
    flips = tokens = len_flips = 0
    worst, nearest = 0.0, 1.0
    for _ in range(20):
        B, Tx = 10, 48
        ids = torch.randint(1, 178, (B, Tx), generator=g)
        xl = torch.randint(8, Tx + 1, (B,), generator=g)
        sid = torch.randint(0, 109, (B,), generator=g)
        spk_c = sds[0]["spk_emb.weight"][sid]
        with torch.inference_mode():
            mu_o, logw_o, x_mask = O.text_encoder(sds[0], ids, xl, spk_c)                  # CPU oracle
        mu_d, logw_d = model.engine.text_encoder(ids.cuda(), xl.cuda(), spk_c.cuda())
        w_o = torch.exp(logw_o) * x_mask
        wd, wo = torch.ceil(torch.exp(logw_d.cpu()) * x_mask), torch.ceil(w_o)
        flips += int((wd != wo).sum())
        tokens += int(x_mask.sum())
        worst = max(worst, float(((logw_d.cpu() - logw_o) * x_mask).abs().max()))
        frac = (w_o - torch.floor(w_o))[x_mask.bool()]
        nearest = min(nearest, float(torch.minimum(frac, 1 - frac).min()))
        for scale in (1.0, 0.8):
            y_o = torch.clamp_min(torch.sum(wo * scale, [1, 2]), 1).long()
            y_d = model._durations(ids, xl, sid, scale)[5].cpu()
            len_flips += int((y_o != y_d).sum())
    print(f"duration flips vs the CPU oracle: {flips} of {tokens} tokens, {len_flips} of 400 mel_lengths over 200 sequences; "
          f"worst |dlogw| {worst:.2e}; nearest exp(logw) to an integer {nearest:.2e}")
    assert worst <= 1e-4
    assert flips == 0 and len_flips == 0, (flips, tokens, len_flips)


# ---------------------------------------------------------------------------------------------------------------------
# robustness
# ---------------------------------------------------------------------------------------------------------------------
def test_two_handles_on_two_host_threads(sds):
    """include/emojivoice.h: different handles are independent.  Two host threads drive two handles (a k = 11 / halo 50
    vocoder and a halo-2 decoder: the launch state that used to be process-global) on two streams at the same time; the
    results must equal the serial run."""
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from emojivoice_amd.matcha_tts import MatchaTTS

    m = MatchaTTS(sds[0], device=DEV)
    v = Generator(AttrDict(v1)).to(DEV)
    v.load_state_dict(sds[1])
    g = torch.Generator().manual_seed(8)
    B, Tp = 2, 64
    mu = torch.randn(B, 80, Tp, generator=g).cuda()
    z = (torch.randn(B, 80, Tp, generator=g) * 0.667).cuda()
    lengths = torch.tensor([64, 41], dtype=torch.int32).cuda()
    spk = m._sd["spk_emb.weight"][torch.tensor([3, 77]).cuda()]
    mel_in = (torch.randn(3, 80, 40, generator=g) * 2 - 5).cuda()
    ref_dec = m.engine.cfm_decode(mu, lengths, spk, z, 3).cpu()
    ref_wav = v(mel_in).cpu()
    torch.cuda.synchronize()
    res, errs = {}, []

    def run_dec():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(6):
                    res["dec"] = m.engine.cfm_decode(mu, lengths, spk, z, 3)
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    def run_voc():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(6):
                    res["wav"] = v(mel_in)
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=run_dec), threading.Thread(target=run_voc)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert torch.equal(res["dec"].cpu(), ref_dec)
    assert torch.equal(res["wav"].cpu(), ref_wav)


def test_token_id_out_of_range_raises(model):
    """nn.Embedding raises IndexError for an id outside the table (text_encoder.py:395); so does the device stage, at the
    path's first synchronisation point.  An out-of-range id in the PADDING of a shorter utterance is never looked at."""
    ids = torch.tensor([[5, 9, 178 + 40, 3]]).cuda()
    with pytest.raises(IndexError):
        model.synthesise(ids, torch.tensor([4]).cuda(), 2, 0.667, torch.tensor([0]).cuda())
    ok = torch.tensor([[5, 9, 7, 999]]).cuda()                     # bad id beyond x_lengths
    out = model.synthesise(ok, torch.tensor([3]).cuda(), 2, 0.667, torch.tensor([0]).cuda())
    assert bool(torch.isfinite(out["mel"]).all())


def test_denoiser_three_frame_utterance(vocoder, sds):
    """768 samples (3 mel frames) is the shortest input torch.stft's reflect padding accepts, hence the shortest the
    reference's to_waveform can denoise; shorter inputs are an error there and here."""
    from emojivoice_amd._lib import EvLibraryError
    from emojivoice_amd.denoiser import Denoiser
    from emojivoice_amd.hifigan import to_waveform

    den = Denoiser(vocoder, mode="zeros")
    mel = torch.randn(1, 80, 3, generator=torch.Generator().manual_seed(6)) * 2 - 5
    got = to_waveform(mel.cuda(), vocoder, den)
    ref = O.to_waveform(sds[1], W.HIFIGAN_V1, mel, O.denoiser_bias_spec(sds[1], W.HIFIGAN_V1))
    assert tuple(got.shape) == tuple(ref.shape) == (768,)
    assert _linf(got, ref) <= 1e-4
    with pytest.raises(EvLibraryError):
        to_waveform(mel[:, :, :2].cuda(), vocoder, den)           # 512 samples: torch.stft raises in the reference too
    with pytest.raises(RuntimeError):
        O.to_waveform(sds[1], W.HIFIGAN_V1, mel[:, :, :2], O.denoiser_bias_spec(sds[1], W.HIFIGAN_V1))


def test_unsupported_head_count_is_refused():
    from emojivoice_amd._lib import Engine, EvLibraryError

    with pytest.raises(EvLibraryError):
        Engine(0, spk_emb_dim=64, heads=4)


# ---------------------------------------------------------------------------------------------------------------------
# callers either side of the path
# ---------------------------------------------------------------------------------------------------------------------
def test_cli_batched_and_phoneme_input(tmp_path):
    """`--batched` (cli.py:277-317: padded id batch, one speaker per line or the default) and `--phonemes` / `--file_phonemes`
    (IPA text through the reference's symbol table) produce one wav per utterance with wav_len == 256 * mel_len; the batched
    run equals the same utterances synthesised as that batch through the Python surface."""
    import wave

    from emojivoice_amd import cli
    from emojivoice_amd import text as T

    lines = ["həlˈoʊ wˈɜːld|12", "ðə kwˈɪk bɹˈaʊn fˈɑːks.|107", "nˈaɪs"]
    f = tmp_path / "utts.txt"
    f.write_text("\n".join(lines), encoding="utf-8")
    out = tmp_path / "out"
    torch.manual_seed(31)
    cli.cli(["--synthetic", "--file", str(f), "--file_phonemes", "--batched", "--batch_size", "3", "--spk", "5", "--steps", "4",
             "--output_folder", str(out)])
    wavs = sorted(out.glob("*.wav"))
    assert [w.name for w in wavs] == ["utterance_001_speaker_012.wav", "utterance_002_speaker_107.wav", "utterance_003_speaker_005.wav"]
    for w, ln in zip(wavs, lines):
        mel = np.load(str(w)[:-4] + ".npy")
        with wave.open(str(w)) as wf:
            assert (wf.getframerate(), wf.getsampwidth()) == (22050, 3) and wf.getnframes() == 256 * mel.shape[1]
    ids0 = T.process_phonemes(lines[0].split("|")[0])
    assert ids0[0] == 0 and ids0[1] == T._symbol_to_id["h"] and len(ids0) == 2 * len("həlˈoʊ wˈɜːld") + 1
    out2 = tmp_path / "out2"
    cli.cli(["--synthetic", "--phonemes", "nˈaɪs", "--emoji-text", "ok \U0001F914", "--steps", "4", "--output_folder", str(out2)])
    assert [w.name for w in sorted(out2.glob("*.wav"))] == ["utterance_001_speaker_017.wav"]


def test_long_utterance_vs_oracle(model, vocoder, sds):
    """A 17-s utterance (T = 1500: 47 key tiles in attention, GroupNorm slabs beyond the register-resident path, 3 M samples of
    audio) next to a short one."""
    sd, voc_sd = sds
    g = torch.Generator().manual_seed(1500)
    B, Tp = 2, 1500
    mu = torch.randn(B, 80, Tp, generator=g)
    z = torch.randn(B, 80, Tp, generator=g)
    lengths = torch.tensor([1500, 333])
    mask = O.sequence_mask(lengths, Tp).unsqueeze(1).float()
    spk = sd["spk_emb.weight"][torch.tensor([22, 66])]
    ref = O.cfm_decode(sd, mu * mask, mask, 2, 0.667, spk, z=z)
    dec, mel = model.decode((mu * mask).cuda(), lengths.cuda(), 2, 0.667, spk.cuda(), z=z.cuda())
    assert _linf(dec, ref) <= MEL_GATE / 2
    ref_wav = O.hifigan_forward(voc_sd, O.denormalize(ref[:1], sd["mel_mean"], sd["mel_std"]), W.HIFIGAN_V1)
    wav = vocoder(mel[:1])
    assert wav.shape == (1, 1, 256 * Tp) and _rms(wav, ref_wav) <= WAV_RMS_GATE / 10


def test_synthesise_sharded_single_rank_equals_synthesise(model, vocoder):
    """dist.synthesise_sharded with one rank (no process group) is synthesise + to_waveform's clamp for the whole batch."""
    from emojivoice_amd import dist as D

    g = torch.Generator().manual_seed(77)
    ids = torch.randint(1, 178, (3, 21), generator=g).cuda()
    xl = torch.tensor([21, 9, 15]).cuda()
    spks = torch.tensor([12, 107, 0]).cuda()
    torch.manual_seed(5)
    out = D.synthesise_sharded(model, vocoder, ids, xl, 3, 0.667, spks, 0.8)
    torch.manual_seed(5)
    ref = model.synthesise(ids, xl, 3, 0.667, spks, 0.8)
    assert torch.equal(out["mel_lengths"].cpu(), ref["mel_lengths"].cpu()) and out["ranks"] == 1
    assert torch.equal(out["mel"], ref["mel"])                # trimmed to max(y_lengths), as synthesise returns it
    assert out["y_max"] == int(ref["mel_lengths"].max()) and out["Tp"] == (out["y_max"] + 3) // 4 * 4
    assert torch.equal(out["wav"], vocoder(ref["mel"]).clamp(-1, 1))     # = to_waveform(synthesise(...)["mel"])


def test_sharded_two_rank_semantics_on_one_gpu(model, vocoder, sds):
    """The N > 1 path with the REAL engine, world size 2 emulated on one GPU: a ragged batch of 5 is computed as
    rank 0's rows and rank 1's rows, one after the other, through ``dist.shard_durations`` / ``shard_decode`` — each at the
    GLOBAL padded length and on its rows of ONE global noise draw, which is where the padding-coupled numerics bite (GroupNorm
    statistics and attention see padded frames, decoder.py:41-43) — and concatenated the way ``all_gather_rows`` collates the
    blocks.  The result must equal the single-batch run on the same engine and pass the oracle gates."""
    from emojivoice_amd import dist as D

    sd, voc_sd = sds
    g = torch.Generator().manual_seed(314)
    B, Lx = 5, 30
    ids = torch.randint(1, 178, (B, Lx), generator=g)
    xl = torch.tensor([30, 11, 24, 7, 19])                     # rank 1's rows are all shorter than rank 0's longest
    spks = torch.tensor([107, 58, 0, 12, 17])
    with torch.inference_mode():
        probe = O.synthesise(sd, ids, xl, 2, 0.667, spks, 1.0, z=None)
    y_max = int(probe["mel_lengths"].max())
    Tp = O.fix_len_compatibility(y_max)
    z = torch.randn(B, 80, Tp, generator=g)
    with torch.inference_mode():
        ref = O.synthesise(sd, ids, xl, 4, 0.667, spks, 1.0, z=z)
        ref_wav = O.hifigan_forward(voc_sd, ref["mel"], W.HIFIGAN_V1).clamp(-1, 1)
    # single batch on the engine
    one = model.synthesise(ids.cuda(), xl.cuda(), 4, 0.667, spks.cuda(), 1.0, z=z.cuda())
    one_wav = vocoder(one["mel"]).clamp(-1, 1)
    # two ranks, sequentially
    states = [D.shard_durations(model, ids.cuda(), xl.cuda(), spks.cuda(), 1.0, r, 2) for r in range(2)]
    assert [(s.lo, s.hi) for s in states] == [(0, 3), (3, 5)] and all(s.error is None for s in states)
    y_max_g = max(s.y_max_local for s in states)               # what agree_max_length's MAX all-reduce returns
    assert y_max_g == y_max and states[1].y_max_local < y_max  # rank 1 alone would have padded to a shorter Tp
    blocks, mels = zip(*[D.shard_decode(model, vocoder, s, y_max_g, z.cuda(), 4, 0.667) for s in states])
    full = torch.cat(blocks, dim=0)                            # all_gather_rows: blocks in rank order (equal-row padding dropped)
    out = D.collate_blocks(full, Tp, y_max_g, 0, 3, 2, mels[0])
    assert torch.equal(out["mel_lengths"].cpu(), ref["mel_lengths"])
    mel2 = torch.cat(mels, dim=0)
    # (i) the sharded computation equals the single batch on the same engine (tile configurations may differ with the batch size)
    assert _linf(mel2, one["mel"].cpu()) <= 1e-5 and _linf(out["wav"], one_wav.cpu()) <= 5e-5
    # (ii) the oracle gates
    assert _linf(mel2, ref["mel"]) <= MEL_GATE and _rms(out["wav"], ref_wav) <= WAV_RMS_GATE
    assert out["wav"].shape == (B, 1, 256 * y_max)


def test_rccl_backend_single_rank_collectives():
    """The N > 1 path uses torch.distributed's "nccl" backend (= RCCL on ROCm).  Only one GPU is visible here, so this checks
    what can be checked: the backend initialises on this box and the two collectives of the path (MAX all-reduce of the padded
    length, all_gather_into_tensor of the waveform block) run on device tensors in a one-rank group."""
    import os
    import subprocess
    import sys

    code = r'''
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([515], dtype=torch.int64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
w = torch.arange(2 * 1024, dtype=torch.float32, device="cuda").reshape(2, 1, 1024); out = torch.empty_like(w)
dist.all_gather_into_tensor(out, w); dist.barrier(); torch.cuda.synchronize()
assert int(t) == 515 and torch.equal(out, w)
dist.destroy_process_group(); print("rccl ok")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stderr[-2000:]


def test_geometry_changes_leave_no_stale_rows(model, vocoder):
    """A change of (B, T) re-plans the workspace and clears only what the new geometry needs (pad rows of the vocoder's tensors,
    the estimator's part whole): results must not depend on what an earlier, differently shaped call left behind."""
    g = torch.Generator().manual_seed(5)
    dev = model.device

    def voc_call(B, T, seed):
        gg = torch.Generator().manual_seed(seed)
        return vocoder((torch.randn(B, 80, T, generator=gg) * 2 - 5).to(dev)).cpu()

    def cfm_call(B, T, seed):
        gg = torch.Generator().manual_seed(seed)
        mu, z = torch.randn(B, 80, T, generator=gg).to(dev), torch.randn(B, 80, T, generator=gg).to(dev)
        lengths = torch.tensor([T] + [max(1, T - 3 * i) for i in range(1, B)], device=dev)
        spk = model._sd["spk_emb.weight"][torch.arange(B, device=dev)]
        return model.engine.cfm_decode(mu, lengths, spk, z, 2).cpu()

    for call in (voc_call, cfm_call):
        first = {}
        shapes = [(1, 36, 1), (2, 20, 2), (1, 100, 3), (3, 12, 4), (1, 36, 1), (1, 8, 5), (2, 20, 2), (1, 100, 3), (1, 8, 5), (3, 12, 4)]
        for B, T, seed in shapes:                      # every shape is visited twice, with other shapes in between
            out = call(B, T, seed)
            assert torch.isfinite(out).all()
            if (B, T) in first:
                assert torch.equal(out, first[(B, T)]), (call.__name__, B, T)
            else:
                first[(B, T)] = out


def test_cfm_decode_is_stream_capturable(sds):
    """ev_cfm_decode enqueues only kernels and one pinned-memory copy on the caller's stream (no host wait under capture), so a
    serving loop may capture a decode of a fixed (B, Tp) in a HIP graph; the replay reproduces the direct call bit for bit — also
    after eager calls with OTHER step counts on the same handle (the captured call's time embeddings sit in pinned memory the
    handle never reuses), and the handle refuses what would pull memory from under the graph (another shape)."""
    from emojivoice_amd._lib import EvLibraryError
    from emojivoice_amd.matcha_tts import MatchaTTS

    model = MatchaTTS(sds[0], device=DEV)                    # its own handle: a captured call binds the handle to its shape
    dev = model.device
    g = torch.Generator().manual_seed(11)
    B, T = 1, 60
    mu, z = torch.randn(B, 80, T, generator=g).to(dev), torch.randn(B, 80, T, generator=g).to(dev)
    lengths = torch.tensor([T], device=dev)
    spk = model._sd["spk_emb.weight"][torch.tensor([4], device=dev)]
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        for _ in range(2):                                    # first calls at this shape allocate the workspace
            ref = model.engine.cfm_decode(mu, lengths, spk, z, 3)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = model.engine.cfm_decode(mu, lengths, spk, z, 3)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize(dev)
        assert torch.equal(out, ref)
        # eager calls with other step counts (other time grids in the handle's staging ring, twice: both ring slots) ...
        ref5 = model.engine.cfm_decode(mu, lengths, spk, z, 5)
        ref7 = model.engine.cfm_decode(mu, lengths, spk, z, 7)
        assert not torch.equal(ref5, ref) and not torch.equal(ref7, ref5)
        out.zero_()
        graph.replay()                                        # ... must not change what the graph computes
        torch.cuda.synchronize(dev)
        assert torch.equal(out, ref)
        # another (smaller) shape is served eagerly beside the graph (ABI 4: every call of a handle that holds graphs re-zeroes its own plan) ...
        small = model.engine.cfm_decode(mu[:, :, :40].contiguous(), torch.tensor([40], device=dev), spk, z[:, :, :40].contiguous(), 3)
        assert bool(torch.isfinite(small).all())
        out.zero_()
        graph.replay()                                        # ... and the graph still computes its own result afterwards
        torch.cuda.synchronize(dev)
        assert torch.equal(out, ref)
        with pytest.raises(EvLibraryError, match="captur"):   # a shape beyond the workspace would replace the arena under the graph
            big = 4 * T
            model.engine.cfm_decode(torch.randn(B, 80, big, device=dev), torch.tensor([big], device=dev), spk, torch.randn(B, 80, big, device=dev), 3)
        with pytest.raises(EvLibraryError, match="captured"):   # more Euler steps than the workspace is planned for
            model.engine.cfm_decode(mu, lengths, spk, z, 100)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize(dev)
        assert torch.equal(out, ref)
    model.engine.close()


def test_decode_graphs_replay_bit_equal_across_lengths(sds):
    """``MatchaTTS.enable_decode_graphs``: one captured ``ev_cfm_decode2`` per padded length on ONE handle.
    Replays must equal the eager calls bit for bit — also after calls (captured and eager) at other lengths in between, in any order —
    and a cache hit must not capture again."""
    from emojivoice_amd.matcha_tts import MatchaTTS

    eager = MatchaTTS(sds[0], device=DEV)
    model = MatchaTTS(sds[0], device=DEV)
    for m in (eager, model):
        m.warmup(max_frames=400, max_tokens=300)
    dg = model.enable_decode_graphs()
    g = torch.Generator().manual_seed(21)
    cases = {}
    for Tp in (64, 200, 132, 396):
        mu = torch.randn(1, 80, Tp, generator=g).to(DEV)
        z = torch.randn(1, 80, Tp, generator=g).to(DEV)
        lengths = torch.tensor([Tp - 3], device=DEV)
        spk = model._sd["spk_emb.weight"][torch.tensor([Tp % 109], device=DEV)]
        ref = eager.decode(mu, lengths, 10, 0.667, spk, z=z)
        cases[Tp] = (mu, z, lengths, spk, ref)
    order = [64, 200, 64, 132, 396, 200, 132, 64, 396]
    for i, Tp in enumerate(order):
        mu, z, lengths, spk, ref = cases[Tp]
        dec, mel = model.decode(mu, lengths, 10, 0.667, spk, z=z)
        torch.cuda.synchronize()
        assert torch.equal(dec, ref[0]) and torch.equal(mel, ref[1]), (i, Tp)
        if i == 4:                                             # an eager call on the SAME handle between replays (another length, another step count)
            model.decode_graphs, keep = None, model.decode_graphs
            e = model.decode(cases[64][0], cases[64][2], 4, 0.667, cases[64][3], z=cases[64][1])
            assert bool(torch.isfinite(e[1]).all())
            model.decode_graphs = keep
    assert dg.captures == 4 and dg.hits == len(order) - 4 and dg.fallbacks == 0, (dg.captures, dg.hits, dg.fallbacks)
    for m in (eager, model):
        m.engine.close()


def test_reserved_handles_do_not_allocate_on_the_request_path(sds):
    """``MatchaTTS.warmup(max_frames=)`` / ``Generator.warmup(max_frames=)`` (ev_reserve) size the workspace, the text-encoder and
    denoiser scratch, the pinned staging ring and the vocoder's side streams once: 20 requests of lengths never seen before (all
    within the reservation) must not move ``ev_alloc_count``, and must still be right (one compared with an unreserved handle).
    Unreserved handles grow geometrically: far fewer allocations than new maxima."""
    from emojivoice_amd import streaming as S
    from emojivoice_amd.denoiser import Denoiser
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from emojivoice_amd.matcha_tts import MatchaTTS

    def pair():
        m = MatchaTTS(sds[0], device=DEV)
        v = Generator(AttrDict(v1)).to(DEV)
        v.load_state_dict(sds[1])
        return m, v

    m, v = pair()
    m.rng = "device"
    den = Denoiser(v, mode="zeros")
    m.warmup(max_frames=1000, max_tokens=700)
    v.warmup(max_frames=1000)
    tts = S.EmojiTTS(m, v, den, text_to_ids=S.table_front_end)
    tts.respond("warm up \U0001F642")
    torch.cuda.synchronize()
    a_m, a_v = m.engine.alloc_count(), v.engine.alloc_count()
    words = "the quick brown fox jumps over the lazy dog and runs far away from here ".split()
    lens = set()
    for i in range(20):
        txt = " ".join(words[(i * 3 + k) % len(words)] for k in range(3 + 2 * i))
        out = tts.respond(txt + " \U0001F60D")
        lens.add(int(out["mel_lengths"][0]))
        assert bool(torch.isfinite(out["waveform"]).all())
    torch.cuda.synchronize()
    assert len(lens) >= 15 and max(lens) <= 1000
    assert (m.engine.alloc_count(), v.engine.alloc_count()) == (a_m, a_v), "a reserved handle allocated on the request path"
    # the same increasing lengths on fresh, unreserved handles: geometric growth, not one allocation per new maximum
    m2, v2 = pair()
    m2.rng = "device"
    tts2 = S.EmojiTTS(m2, v2, Denoiser(v2, mode="zeros"), text_to_ids=S.table_front_end)
    for i in range(20):
        txt = " ".join(words[(i * 3 + k) % len(words)] for k in range(3 + 2 * i))
        tts2.respond(txt + " \U0001F60D")
    torch.cuda.synchronize()
    assert 0 < m2.engine.alloc_count() <= 16 and 0 < v2.engine.alloc_count() <= 16, (m2.engine.alloc_count(), v2.engine.alloc_count())
    for o in (m, v, m2, v2):
        o.engine.close()



def test_reserved_vocoder_mid_size_call_does_not_allocate(sds):
    """the balanced builds' hand-off area (128 MiB + a synchronous memset) used to be allocated by the FIRST vocoder call
    that picked a balanced build (a mid-size batch such as 8 x 516 frames) — on the request path, uncounted, and fatal under stream
    capture.  It now comes with ``ev_load_vocoder`` / ``ev_reserve``: a reserved vocoder handle's first 8 x 516 call must leave
    ``ev_alloc_count`` where it was (the 8-frame warm-up never reaches a balanced build, so it cannot hide the allocation)."""
    from emojivoice_amd.hifigan import AttrDict, Generator, v1

    v = Generator(AttrDict(v1)).to(DEV)
    v.load_state_dict(sds[1])
    v.warmup(max_frames=516, batch=8)
    torch.cuda.synchronize()
    a0 = v.engine.alloc_count()
    mel = torch.randn(8, 80, 516, generator=torch.Generator().manual_seed(3)).to(DEV) * 2 - 5
    wav = v(mel)
    torch.cuda.synchronize()
    assert wav.shape == (8, 1, 516 * 256) and bool(torch.isfinite(wav).all())
    assert v.engine.alloc_count() == a0, "the first mid-size vocoder call allocated on the request path"
    v.engine.close()
