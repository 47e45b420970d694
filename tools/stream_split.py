#!/usr/bin/env python3
"""Where one streaming utterance (feel_me.py loop, B = 1) spends its time: each stage of EmojiTTS.respond timed on its own
with a device synchronisation after it (so the sum is larger than the pipelined end-to-end latency printed last)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import streaming as S, weights as W  # noqa: E402
from emojivoice_amd.denoiser import Denoiser  # noqa: E402
from emojivoice_amd.hifigan import AttrDict, Generator, v1  # noqa: E402
from emojivoice_amd.emoji import parse_response  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402

dev = torch.device("cuda", 0)
model = MatchaTTS(W.synthetic_matcha_state(), device=dev)
voc = Generator(AttrDict(v1)).to(dev)
voc.load_state_dict(W.synthetic_hifigan_state())
den = Denoiser(voc, mode="zeros")
model.rng = os.environ.get("RNG", "cpu")
tts = S.EmojiTTS(model, voc, den, text_to_ids=S.table_front_end)
text = ("the quick brown fox jumps over the lazy dog and keeps running through the quiet meadow " * 3)[: int(os.environ.get("CHARS", "150"))] + " \U0001F600"
sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731


def stage(acc, name, fn):
    sync()
    t0 = time.perf_counter()
    r = fn()
    sync()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r


CLEAN, SPK = parse_response(text)          # the utterance respond() synthesises: same text, same voice
CLEAN = CLEAN.strip()
tts.respond(text)
N = 20
sync()
ts = []
for _ in range(N):
    t0 = time.perf_counter(); tts.respond(text); ts.append((time.perf_counter() - t0) * 1e3)
print("respond() before the staged loop, ms:", " ".join(f"{t:.2f}" for t in ts))
acc = {}
with torch.inference_mode():
    for _ in range(N):
        p = stage(acc, "process_text", lambda: tts.process_text(CLEAN))
        x, xl = p["x"], p["x_lengths"]
        spks = torch.tensor([SPK], device=dev)
        d = stage(acc, "durations (text encoder + host sum)", lambda: model._durations(x, xl, spks, 0.8))
        spk, mu_x, w_ceil, x_mask, xl2, yl, ymax = d
        stage(acc, "text_encoder_status", lambda: model.engine.text_encoder_status())
        from emojivoice_amd.matcha_tts import fix_len_compatibility
        ym = fix_len_compatibility(ymax)
        mu_y, attn = stage(acc, "align", lambda: model.engine.align(w_ceil, mu_x, xl2, yl, ym))
        z = stage(acc, "draw_noise", lambda: model.draw_noise(1, ym))
        dec, mel = stage(acc, "cfm decode", lambda: model.decode(mu_y, yl, 10, 0.667, spk, z=z))
        wav = stage(acc, "vocoder", lambda: voc(mel[:, :, :ymax]))
        wav = stage(acc, "clamp", lambda: wav.clamp(-1, 1))
        dn = stage(acc, "denoiser", lambda: den(wav.squeeze(), strength=0.00025))
        stage(acc, "to host", lambda: dn.cpu().squeeze())
    print(f"frames {ymax}, tokens {int(xl[0])}")
    for k, v in acc.items():
        print(f"  {k:38s} {v / N:7.3f} ms")
    print(f"  {'sum of stages':38s} {sum(acc.values()) / N:7.3f} ms")
    sync()
    ts = []
    for _ in range(N):
        t0 = time.perf_counter(); tts.respond(text); ts.append((time.perf_counter() - t0) * 1e3)
    print("respond() after, ms:", " ".join(f"{t:.2f}" for t in ts))
    print(f"  {'respond() end to end':38s} {sum(ts) / N:7.3f} ms")
from emojivoice_amd.matcha_tts import fix_len_compatibility as _flc  # noqa: E402


def manual(sync_points):
    """respond() restated call by call, with an optional synchronisation after the named stages"""
    def sp(name):
        if name in sync_points:
            sync()
    with torch.inference_mode():
        p = tts.process_text(CLEAN)
        d = model._durations(p["x"], p["x_lengths"], torch.tensor([SPK], device=dev), 0.8)
        spk, mu_x, w_ceil, x_mask, xl2, yl, ymax = d
        ym = _flc(ymax)
        mu_y, attn = model.engine.align(w_ceil, mu_x, xl2, yl, ym); sp("align")
        z = model.draw_noise(1, ym); sp("noise")
        dec, mel = model.decode(mu_y, yl, 10, 0.667, spk, z=z); sp("cfm")
        wav = voc(mel[:, :, :ymax]); sp("voc")
        wav = wav.clamp(-1, 1)
        return den(wav.squeeze(), strength=0.00025).cpu().squeeze()


for pts in ((), ("cfm",), ("cfm", "voc"), ("align", "noise", "cfm", "voc")):
    manual(pts)
    sync()
    t0 = time.perf_counter()
    for _ in range(N):
        manual(pts)
    print(f"  manual pipeline, sync after {pts}: {(time.perf_counter() - t0) * 1e3 / N:7.3f} ms")

# host timeline of ONE pipelined respond(): when each call starts / returns on the host (no synchronisation added)
log = []


evs = []


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        r = f(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        evs.append((label, e0, e1))
        log.append((label, t0, time.perf_counter()))
        return r
    setattr(obj, name, g)


wrap(tts, "process_text", "process_text")
wrap(model, "_durations", "_durations")
wrap(model.engine, "align", "align")
wrap(model, "draw_noise", "draw_noise")
wrap(model.engine, "cfm_decode", "engine.cfm_decode (enqueue)")
wrap(voc, "_sync_engine", "voc._sync_engine")
voc.__class__.__call__ = lambda self, x: self.forward(x)
wrap(voc, "forward", "voc.forward (enqueue)")
den.__class__.__call__ = lambda self, a, strength=0.0005: self.forward(a, strength)
wrap(den, "forward", "den.forward (enqueue)")
sync()
T0 = time.perf_counter()
out = tts.respond(text)
T1 = time.perf_counter()
for label, a, b in log:
    print(f"  {label:32s} start {1e3 * (a - T0):7.3f}  end {1e3 * (b - T0):7.3f}  ({1e3 * (b - a):6.3f} ms)")
print(f"  respond() returned at {1e3 * (T1 - T0):7.3f} ms")
sync()
base = evs[0][1]
for label, e0, e1 in evs:
    print(f"  GPU {label:32s} start {base.elapsed_time(e0):7.3f}  end {base.elapsed_time(e1):7.3f}")
for o in (voc, model, den):
    e = getattr(o, "engine", None)
    if e is not None:
        e.close()
