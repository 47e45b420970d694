#!/usr/bin/env python3
"""Probe: one batch-1 CFM decode captured in a HIP graph (torch.cuda.graph around the C-ABI call) and replayed, against the
normal launch-by-launch call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W  # noqa: E402
from emojivoice_amd.matcha_tts import MatchaTTS  # noqa: E402

dev = torch.device("cuda", 0)
m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
g = torch.Generator().manual_seed(0)
for T in (128, 516, 860):
    mu = torch.randn(1, 80, T, generator=g).to(dev)
    z = torch.randn(1, 80, T, generator=g).to(dev)
    lengths = torch.tensor([T]).to(dev)
    spk = m._sd["spk_emb.weight"][torch.tensor([3]).to(dev)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ref = m.engine.cfm_decode(mu, lengths, spk, z, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m.engine.cfm_decode(mu, lengths, spk, z, 10)
        torch.cuda.synchronize()
        normal = (time.perf_counter() - t0) * 100
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                out = m.engine.cfm_decode(mu, lengths, spk, z, 10)
            gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                gr.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) * 100
            print(f"T={T}: normal {normal:.2f} ms  graph replay {graph:.2f} ms  max|diff| {float((out - ref).abs().max()):.1e}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"T={T}: normal {normal:.2f} ms  capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
            break
m.engine.close()
