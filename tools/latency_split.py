#!/usr/bin/env python3
"""Batch-1 latency split: host enqueue time vs GPU span for the CFM decode and the vocoder (is the path launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
sd, voc_sd, model, voc = bench.build_models(dev)
for T in (128, 256, 516, 860):
    g = torch.Generator().manual_seed(T)
    mu = torch.randn(1, 80, T, generator=g).to(dev); z = (torch.randn(1, 80, T, generator=g) * 0.667).to(dev)
    lengths = torch.tensor([T], device=dev); spk = model._sd["spk_emb.weight"][torch.tensor([12], device=dev)]
    for name, fn in (("cfm", lambda: model.engine.cfm_decode(mu, lengths, spk, z, 10, model.mel_std, model.mel_mean)),
                     ("voc", lambda: voc(mu))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        enq, tot = [], []
        for _ in range(10):
            t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            enq.append(t1 - t0); tot.append(t2 - t0)
        print(f"T={T:4d} {name}: enqueue {1e3*sorted(enq)[5]:.2f} ms  total {1e3*sorted(tot)[5]:.2f} ms", flush=True)
model.engine.close(); voc.engine.close()
