#!/usr/bin/env python3
"""Experiment: TWO batch pipelines in flight (two engine pairs, four streams), batches submitted alternately — is the GPU better
filled with two vocoders + two decodes co-scheduled than with one of each?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W  # noqa: E402
from emojivoice_amd.hifigan import AttrDict, Generator, v1  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402
from emojivoice_amd.pipeline import BatchPipeline  # noqa: E402

B, T, K = int(os.environ.get("B", "64")), 516, int(os.environ.get("K", "8"))
NP = int(os.environ.get("NP", "2"))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
pipes = []
for _ in range(NP):
    m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
    voc = Generator(AttrDict(v1)).to(dev)
    voc.load_state_dict(W.synthetic_hifigan_state())
    pipes.append((m, voc, BatchPipeline(m, voc)))
mu = torch.randn(B, 80, T, generator=g).to(dev)
z = torch.randn(B, 80, T, generator=g).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32).to(dev)
spk = pipes[0][0]._sd["spk_emb.weight"][torch.arange(B, device=dev) % 109]


def run(n_pipes):
    outs = []
    for i in range(K):
        m, voc, bp = pipes[i % n_pipes]
        outs.append(bp.submit(mu, lengths, spk, z, 10))
    for _, _, bp in pipes:
        bp.synchronize()
    return outs


for name, n in (("one pipeline", 1), ("two pipelines", NP), ("one pipeline", 1), ("two pipelines", NP)):
    run(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o = run(n)
    dt = time.perf_counter() - t0
    print(f"{name}: {dt / K * 1e3:.2f} ms per batch of {B}, checksum {float(o[-1].double().abs().sum()):.6f}", flush=True)
for m, voc, _ in pipes:
    m.engine.close()
    voc.engine.close()
