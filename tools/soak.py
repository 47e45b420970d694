#!/usr/bin/env python3
"""Sustained run of the config-2 pipeline (two pipelines in flight): ms per step over time and the balanced grids' hand-off statistics
(launches, arrivals, waits that ran out -> recompute path) of every engine.    python tools/soak.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from emojivoice_amd.pipeline import PipelineGroup
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
sd, voc_sd, m1, v1 = bench.build_models(dev)
_, _, m2, v2 = bench.build_models(dev)
B, T = 64, 516
g = torch.Generator().manual_seed(1234)
mu = torch.randn(B, 80, T, generator=g).to(dev); z = (torch.randn(B, 80, T, generator=g) * 0.667).to(dev)
lengths = torch.full((B,), T, dtype=torch.int32, device=dev)
spk = m1._sd["spk_emb.weight"][torch.randint(0, 11, (B,), generator=g).to(dev)]
pipe = PipelineGroup([(m1, v1), (m2, v2)])
for _ in range(4):
    pipe.submit(mu, lengths, spk, z, 10)
pipe.synchronize()
t0 = time.perf_counter(); marks = []
for i in range(steps):
    pipe.submit(mu, lengths, spk, z, 10)
    if (i + 1) % 10 == 0:
        pipe.synchronize(); marks.append(time.perf_counter())
pipe.synchronize()
prev = t0
for k, t in enumerate(marks):
    print(f"steps {10 * k + 1:3d}..{10 * k + 10:3d}: {(t - prev) / 10 * 1e3:7.2f} ms per step", flush=True); prev = t
for name, e in (("decode 1", m1.engine), ("vocoder 1", v1.engine), ("decode 2", m2.engine), ("vocoder 2", v2.engine)):
    print(name, "balanced launches / arrivals pending / waits that ran out:", e.sk_stats())
pipe.close()
