#!/usr/bin/env python3
"""Experiment: software-pipeline consecutive batches — CFM of batch i+1 on one stream while HiFi-GAN of batch i runs on another."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd import weights as W
from emojivoice_amd.hifigan import AttrDict, Generator, v1
from emojivoice_amd.matcha_tts import MatchaTTS

B, T, K = 64, 516, 6
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
m = MatchaTTS(W.synthetic_matcha_state(), device=dev)
voc = Generator(AttrDict(v1)).to(dev); voc.load_state_dict(W.synthetic_hifigan_state())
mu = torch.randn(B, 80, T, generator=g).to(dev); z = torch.randn(B, 80, T, generator=g).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32).to(dev)
spk = m._sd["spk_emb.weight"][torch.arange(B, device=dev) % 109]

def seq():
    outs = []
    for _ in range(K):
        dec = m.engine.cfm_decode(mu, lengths, spk, z, 10, m.mel_std, m.mel_mean)
        outs.append(voc(dec))
    torch.cuda.synchronize()
    return outs

PA, PB = int(os.environ.get('PA', '0')), int(os.environ.get('PB', '0'))
sA, sB = torch.cuda.Stream(priority=PA), torch.cuda.Stream(priority=PB)
print('priorities', PA, PB)
def pipe():
    outs = []
    for _ in range(K):
        with torch.cuda.stream(sA):
            dec = m.engine.cfm_decode(mu, lengths, spk, z, 10, m.mel_std, m.mel_mean)
            ev = torch.cuda.Event(); ev.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev)
            dec.record_stream(sB)
            outs.append(voc(dec))
    torch.cuda.synchronize()
    return outs

for name, fn in (("sequential", seq), ("pipelined", pipe), ("sequential", seq), ("pipelined", pipe)):
    fn()
    t0 = time.perf_counter(); o = fn(); dt = time.perf_counter() - t0
    print(f"{name}: {dt / K * 1e3:.2f} ms per batch, checksum {float(o[-1].double().abs().sum()):.6f}", flush=True)
