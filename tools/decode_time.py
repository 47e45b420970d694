#!/usr/bin/env python3
"""Wall time of the batch-64 CFM decode (10 steps, T = 516), median of 5: the A/B meter for switches that shape_profile.py does not time
(GroupNorm kernels, launch gaps).    EV_GN_THREADS=512 python tools/decode_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
sd, voc_sd, model, voc = bench.build_models(dev)
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 516
mu, z, spk_ids, lengths = bench.make_inputs(B, T, 0, B, dev)
spk = model._sd["spk_emb.weight"][spk_ids]
for _ in range(2):
    model.engine.cfm_decode(mu, lengths, spk, z, 10)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); model.engine.cfm_decode(mu, lengths, spk, z, 10); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"B={B} cfm_decode ms: median {sorted(ts)[2]:.2f}  all {[round(t, 2) for t in ts]}")
model.engine.close()
