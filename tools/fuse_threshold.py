#!/usr/bin/env python3
"""CFM decode time vs batch with the fused LayerNorm + linear kernels forced on / off (EV_FUSE_MLP_MIN is read at handle creation)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from emojivoice_amd.matcha_tts import MatchaTTS
from emojivoice_amd import weights as W
dev = torch.device("cuda", 0)
sd = W.synthetic_matcha_state()
T = int(os.environ.get("T", "516"))
for B in (1, 2, 4, 8, 16, 32, 64):
    row = []
    for mn in ("1", "1000000"):
        os.environ["EV_FUSE_MLP_MIN"] = mn
        m = MatchaTTS(sd, device=dev)
        g = torch.Generator().manual_seed(B)
        mu = torch.randn(B, 80, T, generator=g).to(dev); z = (torch.randn(B, 80, T, generator=g) * 0.667).to(dev)
        lengths = torch.full((B,), T, device=dev); spk = m._sd["spk_emb.weight"][torch.zeros(B, dtype=torch.long, device=dev)]
        for _ in range(2): m.engine.cfm_decode(mu, lengths, spk, z, 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): m.engine.cfm_decode(mu, lengths, spk, z, 10)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / 3 * 1e3)
        m.engine.close()
    print(f"B={B:3d} T={T}: fused {row[0]:7.2f} ms   separate {row[1]:7.2f} ms   tiles(T)={(B * (T + 4) + 31) // 32}", flush=True)
