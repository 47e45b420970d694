#!/usr/bin/env python3
"""Experiment: split a batch over L independent handles / streams and run them concurrently (tail/head overlap)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1
from emojivoice_amd.matcha_tts import MatchaTTS

B, T = 64, 516
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
mel = (torch.randn(B, 80, T, generator=g) * 2 - 5).to(dev)
mu = torch.randn(B, 80, T, generator=g).to(dev)
z = torch.randn(B, 80, T, generator=g).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32).to(dev)
for L in (1, 2, 4):
    vocs, ms, streams = [], [], []
    for i in range(L):
        v = Generator(AttrDict(v1)).to(dev); v.load_state_dict(W.synthetic_hifigan_state()); vocs.append(v)
        ms.append(MatchaTTS(W.synthetic_matcha_state(), device=dev)); streams.append(torch.cuda.Stream())
    spk = ms[0]._sd["spk_emb.weight"][torch.arange(B, device=dev) % 109]
    n = B // L
    def run(what):
        outs = []
        for i in range(L):
            sl = slice(i * n, (i + 1) * n)
            with torch.cuda.stream(streams[i]):
                if what == "cfm":
                    outs.append(ms[i].engine.cfm_decode(mu[sl].contiguous(), lengths[sl].contiguous(), spk[sl].contiguous(), z[sl].contiguous(), 10))
                else:
                    outs.append(vocs[i](mel[sl].contiguous()))
        torch.cuda.synchronize()
        return torch.cat(outs)
    for what in ("cfm", "voc"):
        ref = run(what)
        t0 = time.perf_counter()
        for _ in range(3):
            run(what)
        dt = (time.perf_counter() - t0) / 3
        print(f"L={L} {what}: {dt * 1e3:8.2f} ms  checksum {float(ref.double().abs().sum()):.6f}", flush=True)
    del vocs, ms
    torch.cuda.empty_cache()
