#!/usr/bin/env python3
"""Accuracy of one deep conv layer (128 -> 128, 7 taps, dilation 3, prologue leaky-relu) under the settings of ev_set_arithmetic against
an fp64 reference on the CPU, for activations of different scale (block scaling of the fp16 form).    python tools/arith_accuracy.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd._lib import Engine
eng = Engine(0)
g = torch.Generator().manual_seed(5)
B, C, T, K, d = 3, 128, 50000, 7, 3
w = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
b = torch.randn(C, generator=g) * 0.1
base = torch.randn(B, C, T, generator=g)
cases = {"x ~ N(0,1)": base, "x ~ N(0,1) * 1e-4": base * 1e-4, "x ~ N(0,1) * 3e3": base * 3e3,
         "rows of very different scale (utterance 0 * 1e-3, 1 * 1, 2 * 1e3)": base * torch.tensor([1e-3, 1.0, 1e3]).view(3, 1, 1),
         "channels of very different scale (x 10^(-3..3))": base * (10.0 ** torch.linspace(-3, 3, C)).view(1, C, 1)}
for name, x in cases.items():
    ref = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), padding=d * (K - 1) // 2, dilation=d)
    print(name)
    for a in (0, 6, 16, 3):
        eng.set_arithmetic(a)
        y = eng.op_conv1d(x.cuda(), w, b, dilation=d, padding=d * (K - 1) // 2, pre_lrelu_slope=0.1).cpu().double()
        # errors relative to the RMS of each utterance's output (utterances may differ in scale by orders of magnitude)
        scale = ref.pow(2).mean(dim=(1, 2), keepdim=True).sqrt()
        e = (y - ref).abs() / scale
        print(f"   arithmetic {a:2d} (build {eng.last_cfg():2d}): max {float(e.max()):.3e}   rms {float(e.pow(2).mean().sqrt()):.3e}")
eng.set_arithmetic(6)
eng.close()
