#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: one warm-up + N HiFi-GAN / CFM calls at a given batch."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W  # noqa: E402
from emojivoice_amd.hifigan import AttrDict, Generator, v1  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--frames", type=int, default=516)
ap.add_argument("--what", default="hifigan", choices=["hifigan", "cfm", "both"])
ap.add_argument("--iters", type=int, default=1)
ap.add_argument("--ode-steps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, T = a.batch, a.frames
g = torch.Generator().manual_seed(0)
if a.what in ("hifigan", "both"):
    voc = Generator(AttrDict(v1)).to(dev)
    voc.load_state_dict(W.synthetic_hifigan_state())
    mel = (torch.randn(B, 80, T, generator=g) * 2 - 5).to(dev)
    for _ in range(a.iters + 1):
        wav = voc(mel)
    torch.cuda.synchronize()
if a.what in ("cfm", "both"):
    m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
    mu = torch.randn(B, 80, T, generator=g).to(dev)
    z = torch.randn(B, 80, T, generator=g).to(dev)
    lengths = torch.full((B,), T).to(dev)
    spk = m._sd["spk_emb.weight"][torch.arange(B, device=dev) % 109]
    for _ in range(a.iters + 1):
        dec = m.engine.cfm_decode(mu, lengths, spk, z, a.ode_steps)
    torch.cuda.synchronize()
for o in (globals().get("voc"), globals().get("m")):       # destroy the native handles while the runtime (and a profiler) is alive
    if o is not None and getattr(o, "engine", None) is not None:
        o.engine.close()
print("done")
