#!/usr/bin/env python3
"""Streaming capacity of one GPU: N independent feel_me.py loops (one host thread, one engine pair and one HIP stream each)
synthesising the same 500-frame utterance back to back — aggregate utterances per second and per-utterance latency vs N."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import streaming as S, weights as W  # noqa: E402
from emojivoice_amd.denoiser import Denoiser  # noqa: E402
from emojivoice_amd.hifigan import AttrDict, Generator, v1  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402

dev = torch.device("cuda", 0)
NMAX = int(os.environ.get("NMAX", "4"))
REPS = int(os.environ.get("REPS", "24"))
text = ("the quick brown fox jumps over the lazy dog and keeps running through the quiet meadow " * 3)[:110] + " \U0001F600"
loops = []
for _ in range(NMAX):
    m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
    m.rng = "device"
    v = Generator(AttrDict(v1)).to(dev)
    v.load_state_dict(W.synthetic_hifigan_state())
    loops.append((S.EmojiTTS(m, v, Denoiser(v, mode="zeros"), text_to_ids=S.table_front_end), m, v, torch.cuda.Stream(device=dev)))


def worker(k, lat, gate, span):
    tts, _, _, stream = loops[k]
    with torch.cuda.stream(stream):
        tts.respond(text)
        gate.wait()                                  # every loop is warm: the timed part starts together
        span[k][0] = time.perf_counter()
        for _ in range(REPS):
            t0 = time.perf_counter()
            out = tts.respond(text)
            lat.append((time.perf_counter() - t0) * 1e3)
        span[k][1] = time.perf_counter()
    worker.frames = int(out["mel_lengths"][0])


for n in range(1, NMAX + 1):
    lats = [[] for _ in range(n)]
    gate, span = threading.Barrier(n), [[0.0, 0.0] for _ in range(n)]
    th = [threading.Thread(target=worker, args=(k, lats[k], gate, span)) for k in range(n)]
    torch.cuda.synchronize()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = max(e for _, e in span) - min(b for b, _ in span)
    allv = sorted(x for l in lats for x in l)
    print(f"{n} loop(s): {n * REPS / dt:7.1f} utterances/s ({worker.frames} frames each), latency p50 {allv[len(allv) // 2]:.2f} ms  max {allv[-1]:.2f} ms", flush=True)
for _, m, v, _ in loops:
    m.engine.close()
    v.engine.close()
