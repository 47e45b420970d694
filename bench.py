#!/usr/bin/env python3
"""EmojiVoice hot-path benchmark (BASELINE.json: audio-seconds per second per GPU and real-time
factor, 10 Euler steps, 22.05 kHz, batch 64 synthetic 6-s utterances).

  python bench.py [--gpus N] [--steps K] [--warmup W]                      config 2 (headline; N > 1 = config 3's per-GPU shape)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --config 4      ODE-step sweep {2,4,10,20,50} at batch 64: latency, mel-MSE vs n = 50 and vs the CPU oracle at the same n
  python bench.py --config 5      streaming feel_me.py loop: 128 mixed-length utterances, all 11 emojis + default, p50 / p99 latency

One "step" (config 2) = CFM decode (n Euler steps of the U-Net estimator) + HiFi-GAN on one batch of B utterances per
GPU (inputs resident in HBM, z given, weights resident), plus — for N > 1 — the single RCCL all-gather that
collates the waveforms.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SR, HOP = 22050, 256
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
ARITH = int(os.environ.get("EV_SPLIT") or 16)   # ev_set_arithmetic preset of the handles this process creates
# products per fp32 multiply-add of the deep layers: 16 -> three fp16 x fp16 products of two block-scaled fp16 pieces (default), 6 / 3 / 9 ->
# bf16 pieces
SPLIT_PRODUCTS = {16: 3, 6: 6, 3: 3, 9: 9}.get(ARITH, 3)
# fp32 in, fp32 out, fp32 accumulation everywhere; the deep layers form each fp32 product from 16-bit pieces on the fp16 / bf16 matrix pipe
# (fp32-grade: tools/arith_accuracy.py, tools/bf16_split_probe.hip), the rest runs on the fp32 MFMA.  EV_SPLIT=0: everything on the fp32 MFMA.
DTYPE = ("f32" if ARITH == 0 else
         "f32 (deep layers: operand = two block-scaled fp16 pieces, product = 3 fp16 products on the fp16 MFMA, f32 accumulate)" if ARITH == 16 else
         f"f32 (deep layers: operand = three bf16 pieces, product = {SPLIT_PRODUCTS} bf16 products on the bf16 MFMA, f32 accumulate)")
PEAK_HBM_GBS = 8000.0
# SURVEY.md §8(d), measured on the reference modules: FLOPs and layer-granular activation bytes per 6-s utterance
ALG_FLOP_PER_AUDIO_S = 62.5e9
ALG_BYTES_PER_AUDIO_S = 520e6
TRAFFIC_JSON = ("profiles/r04_conv_hbm_traffic_pmc.json", "profiles/r03_conv_hbm_traffic_pmc.json", "profiles/r02_conv_hbm_traffic_pmc.json", "profiles/r01_conv_hbm_traffic_pmc.json")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("EV_CPU_THREADS", "32"))))


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def make_inputs(B_global, T, lo, hi, device):
    """Config 2/3 inputs: global tensors from fixed seeds; this rank keeps rows [lo, hi)."""
    from emojivoice_amd.emoji import EMOJI_MAPPING

    mu = torch.randn(B_global, 80, T, generator=torch.Generator().manual_seed(1234))[lo:hi]
    z = torch.randn(B_global, 80, T, generator=torch.Generator().manual_seed(1235))[lo:hi]
    ids = torch.tensor(sorted(EMOJI_MAPPING.values()))
    pick = torch.randint(0, len(ids), (B_global,), generator=torch.Generator().manual_seed(1236))[lo:hi]
    spk_ids = ids[pick]
    lengths = torch.full((hi - lo,), T, dtype=torch.int64)
    return mu.to(device), z.to(device), spk_ids.to(device), lengths.to(device)


def oracle_rows(sd, voc_sd, mu, z0, spk, n_ode):
    """CPU oracle on full-length rows (every length = Tp: rows do not interact): (denormalised mel, waveform).
    ``z0`` is already scaled by the temperature."""
    from emojivoice_amd import weights as W
    from oracle import matcha_oracle as O

    with torch.inference_mode():
        mask = torch.ones(mu.shape[0], 1, mu.shape[2])
        dec = O.solve_euler(sd, z0, mu, mask, n_ode, spk)
        mel = O.denormalize(dec, sd["mel_mean"], sd["mel_std"])
        return mel, O.hifigan_forward(voc_sd, mel, W.HIFIGAN_V1)


def cpu_baseline(sd, voc_sd, mu, z0, spk, n_ode):
    """The CPU oracle (restatement of the reference, pinned by reference-generated goldens) timed on this box's host cores
    on a bounded sample (the first rows) of the same workload: 1 warm-up call, then the median of 3."""
    cores = host_cores()
    torch.set_num_threads(cores)
    sample_b, _, T = mu.shape
    log(f"[bench] cpu baseline: oracle on {cores} host threads ({cpu_model()}), B={sample_b} ...")
    oracle_rows(sd, voc_sd, mu[:1], z0[:1], spk[:1] if spk is not None else None, n_ode)      # warm-up: thread pool, oneDNN primitives
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        mel, wav = oracle_rows(sd, voc_sd, mu, z0, spk, n_ode)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    audio_s = sample_b * T * HOP / SR
    return {"value": round(audio_s / dt, 3), "unit": "audio_s/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"first {sample_b} utterances of the batch (T={T} frames), {n_ode} Euler steps + HiFi-GAN; 1-utterance warm-up, "
                      f"median of 3 calls ({' / '.join(f'{t:.1f}' for t in times)} s wall)"}, mel, wav


def build_models(device):
    from emojivoice_amd import weights as W
    from emojivoice_amd.hifigan import AttrDict, Generator, v1
    from emojivoice_amd.matcha_tts import MatchaTTS

    sd = W.synthetic_matcha_state()
    voc_sd = W.synthetic_hifigan_state()
    model = MatchaTTS(sd, device=device)
    voc = Generator(AttrDict(v1)).to(device)
    voc.load_state_dict(voc_sd)
    return sd, voc_sd, model, voc


def time_text_encoder(model, B, Lx):
    """Text encoder + duration predictor (the stage in front of the timed region; SURVEY §8d asks for it separately):
    device stage (ev_text_encoder) and the plain-torch host-stage variant, B utterances of Lx tokens."""
    g = torch.Generator().manual_seed(77)
    ids = torch.randint(1, model.n_vocab, (B, Lx), generator=g).to(model.device)
    xl = torch.full((B,), Lx, dtype=torch.long, device=model.device)
    spk = model._sd["spk_emb.weight"][torch.randint(0, model.n_spks, (B,), generator=g).to(model.device)] if model.n_spks > 1 else None
    out = {}
    for stage in ("device", "host"):
        model.encoder_stage = stage
        for _ in range(2):
            model.encode(ids, xl, spk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            model.encode(ids, xl, spk)
        torch.cuda.synchronize()
        out[stage] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    model.encoder_stage = "device"
    return {"batch": B, "tokens": Lx, "device_stage_ms": out["device"], "torch_stage_ms": out["host"]}


# ---------------------------------------------------------------------------------------------------------------------
# config 4: ODE-step sweep at batch 64
# ---------------------------------------------------------------------------------------------------------------------
def run_config4(args, device):
    sd, voc_sd, model, voc = build_models(device)
    B, T = args.batch, args.frames
    mu, z, spk_ids, lengths = make_inputs(B, T, 0, B, device)
    spk = model._sd["spk_emb.weight"][spk_ids]
    z = z * 0.667
    s = min(args.cpu_sample, 2, B)
    outs, rows = {}, []
    for n in (50, 2, 4, 10, 20):
        model.engine.cfm_decode(mu, lengths, spk, z, n)                # warm (workspace, code objects)
        torch.cuda.synchronize()
        t_cfm, t_voc = [], []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            dec = model.engine.cfm_decode(mu, lengths, spk, z, n)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            voc(dec * model.mel_std + model.mel_mean)
            torch.cuda.synchronize()
            t_cfm.append(t1 - t0)
            t_voc.append(time.perf_counter() - t1)
        outs[n] = dec
        rows.append((n, sorted(t_cfm)[len(t_cfm) // 2], sorted(t_voc)[len(t_voc) // 2]))
    from oracle import matcha_oracle as O

    res = []
    torch.set_num_threads(host_cores())
    for n, t_cfm, t_voc in sorted(rows):
        with torch.inference_mode():
            ref = O.solve_euler(sd, z[:s].cpu(), mu[:s].cpu(), torch.ones(s, 1, T), n, spk[:s].cpu())   # CPU restatement at the same n
        res.append({"ode_steps": n, "cfm_ms": round(t_cfm * 1e3, 2), "hifigan_ms": round(t_voc * 1e3, 2),
                    "audio_s_per_s": round(B * T * HOP / SR / (t_cfm + t_voc), 1),
                    "mel_mse_vs_50": float(((outs[n] - outs[50]) ** 2).mean()),
                    "mel_mse_vs_cpu_same_n": float(((outs[n][:s].cpu() - ref) ** 2).mean()),
                    "mel_linf_vs_cpu_same_n": float((outs[n][:s].cpu() - ref).abs().max())})
        log(f"[bench] config4 n={n}: {res[-1]}")
    r10 = [r for r in res if r["ode_steps"] == 10][0]
    print(json.dumps({
        "metric": "audio_seconds_per_second", "value": r10["audio_s_per_s"], "unit": "audio_s/s", "n_gpus": 1, "steps": args.steps, "warmup": 1,
        "ms_per_step": round(r10["cfm_ms"] + r10["hifigan_ms"], 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config4: ODE-step sweep {{2,4,10,20,50}} at batch {B} x {T} frames, CFM decode then HiFi-GAN back to back; "
                               "`value` is the n = 10 point", "global_batch": B, "frames": T},
        "sweep": res, "mse_note": f"mel-MSE on decoder outputs; vs_cpu_same_n on the first {s} rows against the CPU oracle at the same n"}), flush=True)
    close_models(model, voc)


# ---------------------------------------------------------------------------------------------------------------------
# config 5: streaming loop
# ---------------------------------------------------------------------------------------------------------------------
_WORDS = ("the of and to in is that it was for on are as with his they at be this from have or by one had not but what all were when we "
          "there can an your which their said if do will each about how up out them then she many some so these would other into has more "
          "her two like him see time could no make than first been its who now people my made over did down only way find use may water "
          "long little very after words called just where most know").split()


def run_config5(args, device):
    """feel_me.py main loop, TTS leg: 128 LLM-style responses of mixed length, emoji cycling through
    the 11 mapped ones + an unmapped one; per utterance: text on the host -> ids -> text encoder -> 10-step CFM at
    SPEAKING_RATE 0.8 -> HiFi-GAN -> clamp -> denoiser -> waveform on the host."""
    import numpy as np

    from emojivoice_amd import streaming as S
    from emojivoice_amd.denoiser import Denoiser
    from emojivoice_amd.emoji import EMOJI_MAPPING

    sd, voc_sd, model, voc = build_models(device)
    den = Denoiser(voc, mode="zeros")
    # the prior sample is drawn on the device, as the reference does when it runs on a GPU (torch.randn_like of a cuda tensor,
    # flow_matching.py:51); MatchaTTS's default ("cpu") reproduces the reference CPU run's stream for seed parity and costs
    # ~1.1 ms of host time per 700-frame utterance (tools/stream_split.py)
    model.rng = "device"
    tts = S.EmojiTTS(model, voc, den, text_to_ids=S.table_front_end)
    g = torch.Generator().manual_seed(4321)
    emojis = list(EMOJI_MAPPING.keys()) + ["\U0001F60A"]
    n_utt = args.utterances
    # utterance lengths T ~ U{86..860} mel frames (1-10 s).  Lengths come out of the duration predictor, so the
    # text length is chosen per utterance from a calibration of frames per token on a probe sentence.
    def make_text(n_chars):
        words, n = [], 0
        while n < n_chars:
            w = _WORDS[int(torch.randint(0, len(_WORDS), (1,), generator=g))]
            words.append(w)
            n += len(w) + 1
        return " ".join(words)[:n_chars]

    # utterance lengths T ~ U{86..860} mel frames (1-10 s).  A length comes out of the duration predictor, so each
    # utterance's text (a prefix of its own seeded word stream) is trimmed in an UNTIMED pre-pass until its mel length is within
    # 10 % of its target; the timed loop below then replays exactly those texts.
    targets = torch.randint(86, 861, (n_utt,), generator=g).tolist()
    resp = []
    for i, tgt in enumerate(targets):
        stream_txt, n_chars = make_text(1500), max(4, tgt // 3)
        for _ in range(6):
            f = float(tts.respond(stream_txt[:n_chars] + " " + emojis[i % len(emojis)])["mel_lengths"][0])
            if abs(f - tgt) <= 0.1 * tgt:
                break
            n_chars = int(min(1500, max(4, round(n_chars * tgt / max(f, 1.0)))))
        resp.append(stream_txt[:n_chars] + " " + emojis[i % len(emojis)])
        if i % 16 == 15:
            log(f"[bench] config5: calibrated {i + 1}/{n_utt} utterance lengths")
    model.warmup()
    voc.warmup()
    tts.respond("warm up " + emojis[0])
    torch.cuda.synchronize()
    lat, audio, frames = [], [], []
    for r in resp:
        t0 = time.perf_counter()
        out = tts.respond(r)                       # ends with the waveform on the host (.cpu())
        lat.append(time.perf_counter() - t0)
        n = int(out["mel_lengths"][0])
        frames.append(n)
        audio.append(n * HOP / SR)
    lat, audio = np.array(lat), np.array(audio)
    p50, p99 = float(np.percentile(lat, 50)) * 1e3, float(np.percentile(lat, 99)) * 1e3
    print(json.dumps({
        "metric": "streaming_tts_latency_p50_ms", "value": round(p50, 2), "unit": "ms", "n_gpus": 1, "steps": n_utt, "warmup": 1,
        "ms_per_step": round(float(lat.mean()) * 1e3, 2), "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"config5: feel_me.py TTS loop, {n_utt} utterances B=1, {min(frames)}-{max(frames)} mel frames, 11 emoji voices + default, "
                               "length_scale 0.8, 10 Euler steps, temperature 0.667 (prior drawn on the device), HiFi-GAN + clamp + denoiser, text on host -> waveform on host"},
        "p50_ms": round(p50, 2), "p99_ms": round(p99, 2), "mean_ms": round(float(lat.mean()) * 1e3, 2), "max_ms": round(float(lat.max()) * 1e3, 2),
        "mean_frames": round(float(np.mean(frames)), 1), "mean_audio_s": round(float(audio.mean()), 2), "mean_rtf": round(float((lat / audio).mean()), 5),
        "x_realtime_stream": round(float(audio.sum() / lat.sum()), 1)}), flush=True)
    close_models(model, voc)


# ---------------------------------------------------------------------------------------------------------------------
# compact config-4 / config-5 records for the default line (so that the driver's own run carries them)
# ---------------------------------------------------------------------------------------------------------------------
def arithmetic_record(voc, mel):
    """HiFi-GAN of the bench batch under the settings of ev_set_arithmetic (16 = default = what `value` is measured with: two block-scaled
    fp16 pieces, three products; 6 = three bf16 pieces, six products; 0 = every layer on the fp32 MFMA; 3 = opt-in fast bf16 setting, NOT
    fp32-grade): ms per call and the waveform difference to the fp32-MFMA result."""
    rec = {"note": "HiFi-GAN alone, serial, same mel; `value` above is measured with 16 unless EV_SPLIT says otherwise; 16 and 6 are fp32-grade, "
                   "3 is an opt-in setting whose products carry ~16 significand bits"}
    outs = {}
    try:
        for a in (0, 16, 6, 3):
            voc.engine.set_arithmetic(a)
            w = voc(mel)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                w = voc(mel)
            torch.cuda.synchronize()
            outs[a] = w
            rec[f"products_{a}"] = {"hifigan_ms": round((time.perf_counter() - t0) / 3 * 1e3, 2)}
    finally:
        voc.engine.set_arithmetic(ARITH)
    for a in (16, 6, 3):
        d = outs[a] - outs[0]
        rec[f"products_{a}"].update({"wav_rms_vs_fp32_mfma": float(d.pow(2).mean().sqrt()), "wav_linf_vs_fp32_mfma": float(d.abs().max())})
    return rec


def config4_record(sd, model, mu, z, spk, lengths, T):
    """ODE-step sweep at the bench batch (flow_matching.py:55-85): CFM decode ms at n in {2, 4, 10, 20, 50}, mel-MSE against the
    n = 50 output, and mel L-inf of ROW 0 against the CPU oracle at the same n."""
    from oracle import matcha_oracle as O

    outs, ms = {}, {}
    for n in (50, 2, 4, 10, 20):
        model.engine.cfm_decode(mu, lengths, spk, z, n)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            outs[n] = model.engine.cfm_decode(mu, lengths, spk, z, n)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ms[n] = sorted(ts)[1]
    torch.set_num_threads(host_cores())
    rec = {"batch": int(mu.shape[0]), "frames": T, "cpu_row": 0, "sweep": []}
    for n in (2, 4, 10, 20, 50):
        with torch.inference_mode():
            ref = O.solve_euler(sd, z[:1].cpu(), mu[:1].cpu(), torch.ones(1, 1, T), n, spk[:1].cpu())
        rec["sweep"].append({"ode_steps": n, "cfm_ms": round(ms[n], 2), "mel_mse_vs_50": float(((outs[n] - outs[50]) ** 2).mean()),
                             "mel_linf_vs_cpu_same_n": float((outs[n][:1].cpu() - ref).abs().max())})
    return rec


def config5_record(model, voc, n_utt=32):
    """feel_me.py TTS loop (feel_me.py:189-203) at B = 1, text on the host -> denoised waveform on the host, twice over n_utt
    mixed-length utterances on handles reserved for 10-s utterances (ev_reserve):
      warm: every text was synthesised once before the timed loop (its length has been planned before);
      cold: texts, hence mel lengths, that this process has never synthesised (each request re-plans the workspace for a new
            length: pad-row zeroing, no allocation — `allocs_during` counts them)."""
    import numpy as np

    from emojivoice_amd import streaming as S
    from emojivoice_amd.denoiser import Denoiser
    from emojivoice_amd.emoji import EMOJI_MAPPING

    rng_was = model.rng
    model.rng = "device"           # the reference draws the prior on the device it runs on (flow_matching.py:51)
    den = Denoiser(voc, mode="zeros")
    model.warmup(max_frames=1200, max_tokens=1000)
    voc.warmup(max_frames=1200)
    tts = S.EmojiTTS(model, voc, den, text_to_ids=S.table_front_end)
    emojis = list(EMOJI_MAPPING.keys()) + ["\U0001F60A"]
    g = torch.Generator().manual_seed(4321)

    def make_text(n_chars, gen):
        words, n = [], 0
        while n < n_chars:
            w = _WORDS[int(torch.randint(0, len(_WORDS), (1,), generator=gen))]
            words.append(w)
            n += len(w) + 1
        return " ".join(words)[:n_chars]

    def run(texts):
        lat, frames = [], []
        for i, t in enumerate(texts):
            t0 = time.perf_counter()
            out = tts.respond(t + " " + emojis[i % len(emojis)])
            lat.append(time.perf_counter() - t0)
            frames.append(int(out["mel_lengths"][0]))
        lat = np.array(lat) * 1e3
        return {"p50_ms": round(float(np.percentile(lat, 50)), 2), "p99_ms": round(float(np.percentile(lat, 99)), 2),
                "mean_ms": round(float(lat.mean()), 2), "mean_frames": round(float(np.mean(frames)), 1),
                "min_frames": int(min(frames)), "max_frames": int(max(frames))}, frames

    tts.respond("warm up " + emojis[0])
    warm_texts = [make_text(int(torch.randint(18, 172, (1,), generator=g)), g) for _ in range(n_utt)]    # ~5 frames per character
    _, seen = run(warm_texts)                                    # untimed pass: every warm length has now been planned once
    torch.cuda.synchronize()
    warm, _ = run(warm_texts)
    g2 = torch.Generator().manual_seed(987)
    cold_texts = [make_text(int(torch.randint(18, 172, (1,), generator=g2)), g2) for _ in range(n_utt)]
    a0 = model.engine.alloc_count() + voc.engine.alloc_count()
    cold, cold_frames = run(cold_texts)
    cold["new_lengths"] = int(sum(1 for f in cold_frames if f not in set(seen)))
    cold["allocs_during"] = int(model.engine.alloc_count() + voc.engine.alloc_count() - a0)
    # the same warm texts with the CFM decode replayed from HIP graphs (MatchaTTS.enable_decode_graphs: one graph per padded length, captured
    # on first use — the untimed pass below — then one hipGraphLaunch instead of ~700 launches per utterance)
    graphs = None
    try:
        dg = model.enable_decode_graphs()
        run(warm_texts)                                          # untimed: captures
        torch.cuda.synchronize()
        graphs, _ = run(warm_texts)
        graphs.update({"graphs_captured": dg.captures, "replays": dg.hits, "eager_fallbacks": dg.fallbacks})
    except Exception as ex:  # noqa: BLE001
        log(f"[bench] config5 graph-replay pass skipped: {type(ex).__name__}: {ex}")
    finally:
        model.decode_graphs = None
    model.rng = rng_was
    return {"utterances": n_utt, "note": "B=1, length_scale 0.8, 10 Euler steps, temperature 0.667, HiFi-GAN + clamp + denoiser; handles reserved for 1200 frames; lengths ~ U{86..860} frames (SURVEY 8d)",
            "warm": warm, "cold_length": cold, "warm_graph_replay": graphs}


# ---------------------------------------------------------------------------------------------------------------------
# config 2 / 3 (default)
# ---------------------------------------------------------------------------------------------------------------------
def close_models(*objs):
    """Destroy the native handles explicitly, while the HIP runtime (and a profiler wrapped around it) is still alive."""
    for o in objs:
        eng = getattr(o, "engine", None)
        if eng is not None:
            eng.close()
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5), help="BASELINE.json config (3 = config 2's per-GPU shape under torchrun)")
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=516, help="mel frames per utterance (516 = 5.99 s)")
    ap.add_argument("--ode-steps", type=int, default=10)
    ap.add_argument("--utterances", type=int, default=128, help="config 5: utterances in the streaming loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="run CFM and HiFi-GAN of each batch back to back on one stream")
    ap.add_argument("--pipelines", type=int, default=2, help="batch pipelines in flight (each its own engine pair and two streams); "
                    "consecutive batches go to them in turn")
    ap.add_argument("--cpu-sample", type=int, default=8)
    ap.add_argument("--no-extras", action="store_true", help="skip the compact config-4 / config-5 records of the default line")
    ap.add_argument("--plain", action="store_true", help="profiling aid: warm-up + K serial steps and nothing else (no roofline / CPU / PCIe legs), "
                    "so that a rocprofv3 --pmc pass sees exactly (W + K) x launches_per_step conv launches")
    args = ap.parse_args()

    from emojivoice_amd import dist as D
    from emojivoice_amd.pipeline import PipelineGroup

    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if args.config == 4:
        return run_config4(args, device)
    if args.config == 5:
        return run_config5(args, device)

    B, T, n_ode = args.batch, args.frames, args.ode_steps
    assert T % 4 == 0
    sd, voc_sd, model, voc = build_models(device)
    lo, hi = D.shard_bounds(B * world, rank, world)
    mu, z, spk_ids, lengths = make_inputs(B * world, T, lo, hi, device)
    spk = model._sd["spk_emb.weight"][spk_ids]
    z = z * 0.667

    def step_local():
        dec = model.engine.cfm_decode(mu, lengths, spk, z, n_ode, model.mel_std, model.mel_mean)   # denormalised mel
        return voc(dec), dec

    # consecutive batches are software-pipelined on two HIP streams (emojivoice_amd/pipeline.py): CFM decode of batch i+1
    # on a high-priority stream while HiFi-GAN of batch i runs; --no-pipeline runs the two stages back to back instead
    # ... and TWO such pipelines are kept in flight (a second engine pair with its own workspace and streams; batches alternate):
    # with two vocoders and up to two decodes co-scheduled the matrix pipes idle less than with one of each (in-run,
    # tools/pipeline2_probe.py: 187.9 -> 185.4 ms per batch; a third pipeline adds nothing).  A batch then takes longer from
    # submission to waveform (see batch_latency_ms for the single-batch latency); `value` is throughput.
    pipe, extra_models = None, []
    if not args.no_pipeline:
        pairs = [(model, voc)]
        for _ in range(max(0, args.pipelines - 1)):
            _, _, m2, v2 = build_models(device)
            extra_models += [m2, v2]
            pairs.append((m2, v2))
        pipe = PipelineGroup(pairs)
        for p_ in pipe.pipes[1:]:                               # one-time workspace allocation of the extra engine pairs (not a step)
            p_.submit(mu, lengths, spk, z, n_ode)
            p_.synchronize()
    pipes = [] if pipe is None else pipe.pipes
    # N > 1: the one collective of the path runs on ONE dedicated stream, ordered behind each batch's vocoder by an event (RCCL
    # keeps per-stream state: alternating the pipelines' streams would serialise its kernels against both)
    gather_stream = torch.cuda.Stream(device=device) if world > 1 else None

    def step():
        """-> (collated waveform of the global batch, this rank's waveform block, this rank's mel)"""
        if pipe is None:
            wav, mel = step_local()
            return (D.all_gather_waveforms(wav, B * world) if world > 1 else wav), wav, mel
        wav, mel = pipe.submit(mu, lengths, spk, z, n_ode, return_mel=True)
        full = wav
        if world > 1:
            gather_stream.wait_event(pipe.last.last_event)
            with torch.cuda.stream(gather_stream):
                wav.record_stream(gather_stream)
                full = D.all_gather_waveforms(wav, B * world)
        return full, wav, mel

    if args.plain:
        for i in range(args.warmup + args.steps):
            step_local()
            torch.cuda.synchronize()
            log(f"[bench] plain step {i + 1}/{args.warmup + args.steps}")
        close_models(model, voc)
        return
    log(f"[bench] rank {rank}/{world}: weights loaded, B={B} T={T}; warmup {args.warmup} ...")
    for _ in range(args.warmup):
        tw = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"[bench] warmup step {time.perf_counter() - tw:.3f} s")
    D.barrier()
    torch.cuda.synchronize()
    sk_engines = [model.engine] + [m.engine for m in extra_models[0::2]]   # the decoders' handles: their balanced launches hand partial tiles over
    sk0 = [e.sk_stats() for e in sk_engines]
    tk0 = [e.sk_taken() for e in sk_engines]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full, wav, mel = step()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    sk1 = [e.sk_stats() for e in sk_engines]
    tk1 = [e.sk_taken() for e in sk_engines]
    handoffs = {"balanced_launches": int(sum(b[0] - a[0] for a, b in zip(sk0, sk1))), "waits_ran_out": int(sum(b[2] - a[2] for a, b in zip(sk0, sk1))),
                "shares_taken_over": int(sum(b - a for a, b in zip(tk0, tk1))),
                "note": "over the timed steps, all pipelines: launches of the balanced persistent builds; hand-off waits that ran out (a contributor that had started did not "
                        "deliver within the spin limit: the owner recomputed its share); shares an owner took over at once because the contributor had not started (not resident)"}
    dt_rank = dt
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    per_rank_ms = [round(dt / args.steps * 1e3, 2)]
    allgather_ms = None
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        # every rank's own time for the K steps (before the closing barrier it is the rank's compute + its share of the gathers)
        trs = torch.zeros(world, dtype=torch.float64, device=device)
        trs[rank] = dt_rank
        torch.distributed.all_reduce(trs)
        per_rank_ms = [round(float(v) / args.steps * 1e3, 2) for v in trs.tolist()]
        # the collective alone: 5 gathers of this rank's last waveform block, nothing else in flight
        torch.cuda.synchronize()
        D.barrier()
        tg = time.perf_counter()
        for _ in range(5):
            D.all_gather_waveforms(wav, B * world)
        torch.cuda.synchronize()
        tga = torch.tensor([(time.perf_counter() - tg) / 5 * 1e3], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tga, op=torch.distributed.ReduceOp.MAX)
        allgather_ms = round(float(tga.item()), 3)
    dt = float(tmax.item())
    log(f"[bench] timed {args.steps} steps in {dt:.3f} s")
    audio_s_total = args.steps * B * world * T * HOP / SR
    value = audio_s_total / dt
    # every rank took part: the collated block must hold B rows per rank, each rank's rows at its offset
    ranks_seen = world
    if world > 1:
        seen = torch.zeros(world, dtype=torch.int64, device=device)
        seen[rank] = 1
        torch.distributed.all_reduce(seen)
        ranks_seen = int((seen > 0).sum())
        assert ranks_seen == world, f"only {ranks_seen} of {world} ranks reported"
        assert tuple(full.shape) == (B * world, 1, T * HOP), tuple(full.shape)
        assert torch.equal(full[lo:hi], wav), "this rank's rows are not at their global offset in the gathered block"
    gathered_shape = list(full.shape)

    if rank == 0:
        # ---- the same workload on the serial schedule (one stream, stages back to back): what the roofline pass below runs on
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(3):
            step_local()
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - ts) / 3 * 1e3
        # ---- roofline of the dominant kernel family (fp32-MFMA implicit-GEMM conv), measured live with HIP events
        # recorded on the launch stream around every conv launch of one extra, untimed-for-`value` step
        for e in (model.engine, voc.engine):
            e.profile_enable(True)
        step_local()                      # rank-local: no collective outside the timed region
        torch.cuda.synchronize()
        sp_c, sp_v = model.engine.profile_read_split(), voc.engine.profile_read_split()
        ms_c, fl_c, n_c = model.engine.profile_read()
        ms_v, fl_v, n_v = voc.engine.profile_read()
        for e in (model.engine, voc.engine):
            e.profile_enable(False)
        conv_ms, conv_fl, conv_n = ms_c + ms_v, fl_c + fl_v, n_c + n_v
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12
        # HBM bytes per conv launch from committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # in separate runs, FETCH doubled per the gfx950 note in MI355X_MICROARCH.md): NOT measured in this run
        traffic = None
        for tj_path in TRAFFIC_JSON:
            try:
                tj = json.load(open(os.path.join(REPO, tj_path)))
                if B == 64 and T == 516 and n_ode == 10:
                    import hashlib
                    hsh = hashlib.sha256()
                    for rel in ("emojivoice_amd/csrc/ev_kernels.h", "emojivoice_amd/csrc/ev_engine.hip", "include/emojivoice.h"):
                        hsh.update(open(os.path.join(REPO, rel), "rb").read())
                    traffic = {"bytes_per_launch": round(tj["hbm_MB_per_launch"] * 1e6), "GB_per_step": round(tj["hbm_GB_per_step"], 1),
                               "measured_in_this_run": False, "source": tj_path + " (builder's earlier rocprofv3 --pmc passes of this command)",
                               "pmc_library_source_sha16": tj.get("library_source_sha16"), "this_library_source_sha16": hsh.hexdigest()[:16],
                               "same_library": tj.get("library_source_sha16") == hsh.hexdigest()[:16]}
                    break
            except Exception:
                pass
        per_gpu = value / world
        # The dominant family = the launches on the bf16 matrix pipe (every fp32 multiply-add as SPLIT_PRODUCTS exact bf16 products,
        # fp32 accumulation): algorithmic FLOP / time against dense-bf16-peak / SPLIT_PRODUCTS, which is the same fraction as
        # executed MFMA FLOP against the bf16 peak.  The launches still on v_mfma_f32_32x32x2_f32 are priced against the fp32 peak.
        sp_ms, sp_fl, sp_n = sp_c[0] + sp_v[0], sp_c[1] + sp_v[1], sp_c[2] + sp_v[2]
        f32_ms, f32_fl, f32_n = conv_ms - sp_ms, conv_fl - sp_fl, conv_n - sp_n
        peak_split = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS      # (the fp16 MFMA has the bf16 MFMA's dense peak)
        if sp_n > 0:
            ach_split = sp_fl / (sp_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": round(ach_split, 2), "peak": round(peak_split, 1), "unit": "TFLOP/s", "frac": round(ach_split / peak_split, 4),
                        "traffic": traffic,
                        "kernel": ("conv_h16_kernel + conv_h16_bal_kernel + resblock_pair_h16_kernel + ln_mlp_h16_kernel + ln_qkv_h16_kernel + attn_out_h16_kernel (fp32 contractions as 3 fp16 "
                                   "products of two block-scaled fp16 pieces per operand on v_mfma_f32_32x32x16_f16 / 16x16x32_f16, fp32 accumulation; the attention's QK^T and PV "
                                   "execute all 4 piece products and are still priced at 3)") if ARITH == 16 else
                                  ("conv_split_kernel + conv_split_bal_kernel + resblock_pair_split_kernel + ln_mlp_split_kernel (fp32 contractions as "
                                   f"{SPLIT_PRODUCTS} bf16 products per element pair on v_mfma_f32_32x32x16_bf16, fp32 accumulation)"),
                        "peak_note": f"2500 TFLOP/s dense 16-bit MFMA / {SPLIT_PRODUCTS} products per fp32 multiply-add; `achieved` counts ALGORITHMIC fp32 FLOP"
                                     ,
                        "executed_tflops": round(ach_split * SPLIT_PRODUCTS, 1), "peak_executed": PEAK_BF16_MFMA_TFLOPS,
                        "sustained_note": "with real operand data the pure 16-bit MFMA loop is power-limited to ~1800 TFLOP/s (1.4 kW, clock 2.4 -> ~1.78 GHz: "
                                          "tools/bf16_split_probe.hip, profiles/r03_clock_and_power_vocoder_loop.txt), i.e. ~600 TFLOP/s of fp32-equivalent work at 3 "
                                          "products, ~300 at 6",
                        "launches_per_step": int(sp_n), "avg_launch_us": round(sp_ms * 1e3 / sp_n, 2), "alg_gflop_per_launch": round(sp_fl / sp_n / 1e9, 3),
                        "ms_per_step": round(sp_ms, 2),
                        "fp32_mfma_family": {"achieved": round(f32_fl / (f32_ms * 1e-3) / 1e12, 2) if f32_n else None, "peak": PEAK_FP32_MFMA_TFLOPS,
                                             "frac": round(f32_fl / (f32_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if f32_n else None,
                                             "kernel": "conv_gemm_kernel (+ attn_out_kernel where the fp16 attention does not apply): layers whose channel counts the split builds do not take",
                                             "launches_per_step": int(f32_n), "ms_per_step": round(f32_ms, 2)}}
        else:
            roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                        "kernel": "conv_gemm_kernel + resblock_pair_kernel + ln_mlp_kernel + attn_out_kernel (fp32 v_mfma_f32_32x32x2_f32 contractions; EV_SPLIT=0)",
                        "launches_per_step": int(conv_n), "avg_launch_us": round(conv_ms * 1e3 / max(conv_n, 1), 2),
                        "alg_gflop_per_launch": round(conv_fl / max(conv_n, 1) / 1e9, 3)}
        roofline.update({"schedule": "serial pass (one stream, stages back to back, see serial_ms_per_step); `value` is the two-stream pipeline",
                         "family_launches_per_step": int(conv_n), "family_tflops": round(achieved, 2),
                         "conv_ms_per_step": round(conv_ms, 2), "conv_ms_cfm": round(ms_c, 2), "conv_ms_hifigan": round(ms_v, 2),
                         "tflops_cfm_convs": round(fl_c / (ms_c * 1e-3) / 1e12, 2), "tflops_hifigan_convs": round(fl_v / (ms_v * 1e-3) / 1e12, 2)})
        path_roof = {"mfma_frac": round(per_gpu * ALG_FLOP_PER_AUDIO_S / (peak_split * 1e12), 4),
                     "fp32_mfma_equiv": round(per_gpu * ALG_FLOP_PER_AUDIO_S / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                     "hbm_frac": round(per_gpu * ALG_BYTES_PER_AUDIO_S / (PEAK_HBM_GBS * 1e9), 4),
                     "note": "whole-path fractions from SURVEY §8(d) per-audio-second work: against the split builds' roof (16-bit MFMA peak / products per multiply-add), against the "
                             "fp32 MFMA peak (> 1 = faster than any exact-fp32-MFMA implementation could be), and against HBM"}
        # stage times of one batch (serial schedule) and the latency a single request sees
        torch.cuda.synchronize()
        tl = time.perf_counter()
        dec_l = model.engine.cfm_decode(mu, lengths, spk, z, n_ode, model.mel_std, model.mel_mean)
        torch.cuda.synchronize()
        tm = time.perf_counter()
        voc(dec_l)
        torch.cuda.synchronize()
        te = time.perf_counter()
        batch_latency_ms = (te - tl) * 1e3
        stage_ms = {"cfm_decode": round((tm - tl) * 1e3, 2), "hifigan": round((te - tm) * 1e3, 2)}
        text_enc = time_text_encoder(model, B, 151)
        # PCIe-inclusive rate (never `value`): the same step plus the D2H copy of the waveform block into pinned host memory
        pcie = None
        try:
            host_wav = torch.empty((B,) + tuple(wav.shape[1:]), dtype=torch.float32, pin_memory=True)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            host_wav.copy_(step_local()[0], non_blocking=True)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - tp
            td = time.perf_counter()
            host_wav.copy_(wav[:B], non_blocking=True)
            torch.cuda.synchronize()
            pcie = {"audio_s_per_s_per_gpu": round(B * T * HOP / SR / dtp, 2), "ms_per_step": round(dtp * 1e3, 2),
                    "d2h_ms": round((time.perf_counter() - td) * 1e3, 3), "d2h_MB": round(host_wav.numel() * 4 / 1e6, 1)}
            del host_wav
        except Exception as ex:  # pinned allocation can be refused on small hosts: the field stays null
            log(f"[bench] pcie-inclusive probe skipped: {ex}")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            s = min(args.cpu_sample, B)
            cpu, ref_mel, ref_wav = cpu_baseline(sd, voc_sd, mu[:s].cpu(), z[:s].cpu(), spk[:s].cpu(), n_ode)
            # parity of the TIMED output itself: the first rows of the last timed step's mel / waveform (B = 64 tile
            # configurations, two-stream schedule) against the oracle on the same rows
            cpu["parity_mel_linf"] = float((mel[:s].cpu() - ref_mel).abs().max())
            cpu["parity_wav_rms"] = float((wav[:s].cpu() - ref_wav).pow(2).mean().sqrt())
            cpu["parity_wav_linf"] = float((wav[:s].cpu() - ref_wav).abs().max())
            cpu["parity_rows"] = f"rows 0..{s - 1} of the last timed step's output (batch {B})"
        # ---- the same step with EVERY product on the exact fp32 MFMA (ev_set_arithmetic 0: v_mfma_f32_32x32x2_f32 = an fmaf chain, bit for bit),
        # on the schedule `value` is measured on: the strict-fp32 throughput beside the headline.  The reference's
        # arithmetic is fp32 (flow_matching.py:32-85, hifigan/models.py:181-197); the headline's products carry 22-23 significand bits.
        fp32 = None
        if ARITH != 0:
            engines = [model.engine, voc.engine] + [m.engine for m in extra_models]
            try:
                for e in engines:
                    e.set_arithmetic(0)
                step(); torch.cuda.synchronize()            # warm-up: code objects of the fp32 builds
                D.barrier()
                tf = time.perf_counter()
                for _ in range(3):
                    _, wav32, mel32 = step()
                torch.cuda.synchronize()
                dtf = time.perf_counter() - tf
                fp32 = {"value": round(3 * B * T * HOP / SR / dtf, 2), "unit": "audio_s/s per GPU", "ms_per_step": round(dtf / 3 * 1e3, 2), "steps": 3,
                        "schedule": "as `value`" if pipe is not None else "serial",
                        "mel_linf_vs_headline_arithmetic": float((mel32 - mel).abs().max()), "wav_rms_vs_headline_arithmetic": float((wav32 - wav).pow(2).mean().sqrt()),
                        "note": "ev_set_arithmetic(0): every contraction on v_mfma_f32_32x32x2_f32 (exact fp32 multiply-add chain); same inputs, same schedule, this rank"}
            except Exception as ex:  # noqa: BLE001 - a side record must not take the headline line down
                log(f"[bench] fp32-MFMA step skipped: {type(ex).__name__}: {ex}")
            finally:
                for e in engines:
                    e.set_arithmetic(ARITH)
        c4 = c5 = arith = None
        if world == 1 and not args.no_extras and B == 64 and T == 516:
            if pipe is not None:
                pipe.close()                                    # (gives the vocoder engine its small-call three-stream fan-out back)
            try:
                arith = arithmetic_record(voc, mel)
            except Exception as ex:  # noqa: BLE001
                log(f"[bench] arithmetic record skipped: {type(ex).__name__}: {ex}")
            try:
                c4 = config4_record(sd, model, mu, z, spk, lengths, T)
                c5 = config5_record(model, voc)
            except Exception as ex:  # noqa: BLE001 - the headline line must not die with a side record
                log(f"[bench] config-4 / config-5 records skipped: {type(ex).__name__}: {ex}")
        out = {
            "metric": "audio_seconds_per_second", "value": round(value, 2), "unit": "audio_s/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"config2: batch {B} x {T}-frame (5.99 s) utterances per GPU, {n_ode} Euler steps + HiFi-GAN V1, 22.05 kHz",
                       "global_batch": B * world, "frames": T, "ode_steps": n_ode, "parallelism": f"dp{world}",
                       "collective": "all_gather(waveforms)" if world > 1 else "none",
                       "arithmetic": "f32 tensors and f32 accumulation throughout; EV_SPLIT=0 -> every product on the f32 MFMA" if ARITH == 0 else
                                     ("f32 tensors and f32 accumulation throughout; deep layers: operand = two fp16 pieces of the operand times a power-of-two block "
                                      "scale (22-23 significand bits), product = h0g0 + h0g1 + h1g0; one conv layer vs fp64 (tools/arith_accuracy.py): rms 3.4e-7 of "
                                      "the output scale against 5.4e-7 for the f32 MFMA; parity_* below are measured on this run's timed output") if ARITH == 16 else
                                     "f32 tensors and f32 accumulation throughout; deep layers: operand = exact sum of 3 bf16 pieces, product = the piece products of "
                                     "weight <= 2 (6) / <= 1 (3); profiles/r03_bf16_split_probe.txt; parity_* below are measured on this run's timed output",
                       "batch_pipeline": "off" if pipe is None else f"cfm(i+1) || hifigan(i) on two streams, {len(pipes)} pipeline(s) in flight",
                       "memory": "off" if pipe is None else f"{len(pipes)} engine pairs resident per GPU, each its own weights (about 0.45 GB: fp32 fragments + bf16 and fp16 piece planes in both MFMA layouts) + workspace "
                                 f"({round(model.engine.workspace_bytes(B, T, 0) / 1e9 + voc.engine.workspace_bytes(B, 0, T) / 1e9, 1)} GB at this shape)"},
            "ranks_seen": ranks_seen, "gathered_shape": gathered_shape, "per_rank_ms_per_step": per_rank_ms, "allgather_ms": allgather_ms,
            "per_gpu_audio_s_per_s": round(per_gpu, 2), "rtf": round(1.0 / per_gpu, 6), "x_realtime_per_gpu": round(per_gpu, 1),
            "serial_ms_per_step": round(serial_ms, 2), "batch_latency_ms": round(batch_latency_ms, 2), "stage_ms": stage_ms,
            "text_encoder": text_enc,
            "value_fp32_mfma": fp32, "balanced_handoffs": handoffs,
            "roofline": roofline, "path_roofline": path_roof, "cpu_baseline": cpu, "pcie_inclusive": pcie,
            "config4": c4, "config5": c5, "arithmetic_settings": arith,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()
    close_models(model, voc, *extra_models)


if __name__ == "__main__":
    main()
