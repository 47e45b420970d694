#!/usr/bin/env python3
"""BASELINE.json configs 4 and 5 on one MI355X (config 2/3 are bench.py, config 1 is tests/test_gpu_parity.py::test_cli...).

  config 4: ODE-step sweep {2,4,10,20,50} at batch 64 — latency and mel-MSE vs the 50-step output
  config 5: streaming loop, 128 mixed-length utterances (T ~ U{86..860} frames, seed 4321), batch 1 each, all 11 emoji
            speakers + default 0 — p50/p99 end-to-end latency (ids on host -> wav on host)
Prints one JSON object per config.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W  # noqa: E402
from emojivoice_amd.emoji import EMOJI_MAPPING  # noqa: E402
from emojivoice_amd.hifigan import AttrDict, Generator, v1  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402

dev = torch.device("cuda", 0)
sd = W.synthetic_matcha_state()
model = MatchaTTS(sd, device=dev)
voc = Generator(AttrDict(v1)).to(dev)
voc.load_state_dict(W.synthetic_hifigan_state())
which = sys.argv[1:] or ["4", "5"]

if "4" in which:
    B, T = 64, 516
    mu = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(1234)).to(dev)
    z = (torch.randn(B, 80, T, generator=torch.Generator().manual_seed(1235)) * 0.667).to(dev)
    ids = torch.tensor(sorted(EMOJI_MAPPING.values()))
    spk = model._sd["spk_emb.weight"][ids[torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(1236))].to(dev)]
    lengths = torch.full((B,), T, device=dev)
    outs, rows = {}, []
    for n in (50, 2, 4, 10, 20, 50):
        model.engine.cfm_decode(mu, lengths, spk, z, n)  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec = model.engine.cfm_decode(mu, lengths, spk, z, n)
        torch.cuda.synchronize()
        t_cfm = time.perf_counter() - t0
        t0 = time.perf_counter()
        wav = voc(dec * model.mel_std + model.mel_mean)
        torch.cuda.synchronize()
        t_voc = time.perf_counter() - t0
        outs[n] = dec
        if len(rows) or n != 50:
            pass
        rows.append((n, t_cfm, t_voc))
    ref = outs[50]
    res = []
    for n, t_cfm, t_voc in rows[1:]:
        mse = float(((outs[n] - ref) ** 2).mean())
        res.append({"ode_steps": n, "cfm_ms": round(t_cfm * 1e3, 2), "hifigan_ms": round(t_voc * 1e3, 2),
                    "audio_s_per_s": round(B * T * 256 / 22050 / (t_cfm + t_voc), 1), "mel_mse_vs_50": mse})
    print(json.dumps({"config": 4, "workload": "batch 64 x 516 frames, ODE-step sweep", "results": res}), flush=True)

if "5" in which:
    g = torch.Generator().manual_seed(4321)
    Ts = torch.randint(86, 861, (128,), generator=g)
    spk_cycle = sorted(EMOJI_MAPPING.values()) + [0]
    lat, audio = [], []
    # warm every distinct Tp once is NOT done: a streaming server sees new lengths all the time
    for i, T in enumerate(Ts.tolist()):
        Tp = (T + 3) // 4 * 4
        Lx = max(8, T // 4)
        ids_h = torch.randint(1, 178, (1, Lx), generator=g)
        mu_h = torch.randn(1, 80, Tp, generator=g)
        spk_h = torch.tensor([spk_cycle[i % len(spk_cycle)]])
        t0 = time.perf_counter()
        x = ids_h.to(dev)
        spk = model._sd["spk_emb.weight"][spk_h.to(dev)]
        model.encode(x, torch.tensor([Lx], device=dev), spk)              # text encoder + duration predictor (its durations are replaced by the target T)
        mu = mu_h.to(dev)
        lengths = torch.tensor([T], device=dev)
        dec, mel = model.decode(mu, lengths, 10, 0.667, spk)
        wav = voc(mel).clamp(-1, 1)[:, :, : T * 256].cpu()
        lat.append(time.perf_counter() - t0)
        audio.append(T * 256 / 22050)
    lat = np.array(lat)
    audio = np.array(audio)
    print(json.dumps({"config": 5, "workload": "128 utterances, B=1, T~U{86..860} frames, ids-on-host -> wav-on-host, 10 ODE steps",
                      "p50_ms": round(float(np.percentile(lat, 50)) * 1e3, 2), "p99_ms": round(float(np.percentile(lat, 99)) * 1e3, 2),
                      "mean_ms": round(float(lat.mean()) * 1e3, 2), "first_call_ms": round(float(lat[0]) * 1e3, 2),
                      "mean_audio_s": round(float(audio.mean()), 2), "mean_rtf": round(float((lat / audio).mean()), 5),
                      "x_realtime_stream": round(float(audio.sum() / lat.sum()), 1)}), flush=True)
