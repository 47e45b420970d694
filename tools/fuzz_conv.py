#!/usr/bin/env python3
"""Randomised operator fuzz of the conv kernels through ev_op_conv1d against torch CPU conv1d / conv_transpose1d:
random (B, Cin, Cout, K, dilation, T) incl. ragged channel counts, for the tile configuration forced by EV_FORCE_CFG
(run once per configuration: the switch is read once per process).
    EV_FORCE_CFG=<cfg> python tools/fuzz_conv.py [n_cases] [seed]"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd._lib import Engine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
eng = Engine(0)
worst = 0.0
for i in range(n_cases):
    B = int(rng.integers(1, 4)); T = int(rng.integers(1, 400))
    Cin = 4 * int(rng.integers(1, 80)); Cout = int(rng.choice([1, 3, 32, 64, 80, 100, 128, 192, 256, 300]))
    K = int(rng.choice([1, 3, 5, 7, 11])); dil = int(rng.choice([1, 1, 2, 3, 5]))
    slope = float(rng.choice([-1.0, 0.1]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    if (K * dil - dil) // 2 > 32:
        continue
    y = eng.op_conv1d(x.cuda(), w, b, dilation=dil, padding=(K * dil - dil) // 2, pre_lrelu_slope=slope).cpu()
    xin = F.leaky_relu(x, slope) if slope >= 0 else x
    ref = F.conv1d(xin, w, b, dilation=dil, padding=(K * dil - dil) // 2)
    err = float((y - ref).abs().max())
    worst = max(worst, err)
    if err > 2e-4:
        print(f"MISMATCH case {i}: B={B} Cin={Cin} Cout={Cout} K={K} dil={dil} T={T} slope={slope} err={err:.3e}")
        sys.exit(1)
for i in range(n_cases // 4):
    B = int(rng.integers(1, 3)); T = int(rng.integers(1, 200))
    Cin = 4 * int(rng.integers(1, 64)); Cout = 4 * int(rng.integers(1, 64))
    s = int(rng.choice([2, 4, 8])); K = 2 * s; p = s // 2
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cin, Cout, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    y = eng.op_conv1d(x.cuda(), w, b, transposed=True, stride=s, padding=p).cpu()
    ref = F.conv_transpose1d(x, w, b, stride=s, padding=p)
    err = float((y - ref).abs().max())
    worst = max(worst, err)
    if err > 2e-4:
        print(f"MISMATCH convT case {i}: B={B} Cin={Cin} Cout={Cout} K={K} s={s} T={T} err={err:.3e}")
        sys.exit(1)
print(f"cfg {os.environ.get('EV_FORCE_CFG', 'auto')}: {n_cases} conv + {n_cases // 4} convT cases ok, worst abs err {worst:.2e}")
