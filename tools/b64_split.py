#!/usr/bin/env python3
"""Config-2 stage times at batch B (default 64, T = 516): CFM decode (10 Euler steps) and HiFi-GAN, each timed on its own with
HIP events over N calls — the in-run A/B harness for kernel changes (EV_LIB_PATH selects the library)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W  # noqa: E402
from emojivoice_amd.hifigan import AttrDict, Generator, v1  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402

B, T, N = int(os.environ.get("B", "64")), int(os.environ.get("T", "516")), int(os.environ.get("N", "5"))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
voc = Generator(AttrDict(v1)).to(dev)
voc.load_state_dict(W.synthetic_hifigan_state())
m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
mel = (torch.randn(B, 80, T, generator=g) * 2 - 5).to(dev)
mu = torch.randn(B, 80, T, generator=g).to(dev)
z = torch.randn(B, 80, T, generator=g).to(dev)
lengths = torch.full((B,), T).to(dev)
spk = m._sd["spk_emb.weight"][torch.arange(B, device=dev) % 109]


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N


c = timed(lambda: m.engine.cfm_decode(mu, lengths, spk, z, 10))
v = timed(lambda: voc(mel))
print(f"B={B} T={T}: cfm {c:.2f} ms  hifigan {v:.2f} ms  sum {c + v:.2f} ms")
m.engine.close()
voc.engine.close()
