#!/usr/bin/env python3
"""Conv-kernel microbenchmark / ablation on the GPU box: per HiFi-GAN level and estimator shape, TFLOP/s of
conv_gemm_kernel for tile configs and ablation bits (1: no X loads, 2: no A loads, 4: no epilogue; 16: per-workgroup
s_memrealtime stamps {start, staging done, K loop done, end}, with 1024 / 2048 moving the second stamp inside the staging of the
small-launch build; 128: no residual; 512: no prologue activation = the estimator's layers, which is what conv_sk32_kernel takes).
EST_B=<batch> sets the batch of the "est" shapes (1 = a streaming decode), SHAPES=<prefixes> selects rows."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd._lib import Engine  # noqa: E402

eng = Engine(0)
fn = eng.lib.ev_dbg_conv_bench
fn.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_float)]
B = int(os.environ.get("B", "16"))
shapes = [  # name, Cin, Cout, K, dil, T, P
    ("L1 k3", 256, 256, 3, 1, 4128, 32), ("L1 k7d3", 256, 256, 7, 3, 4128, 32), ("L1 k11d5", 256, 256, 11, 5, 4128, 32),
    ("L2 k3", 128, 128, 3, 1, 33024, 256), ("L2 k7d3", 128, 128, 7, 3, 33024, 256), ("L2 k11d5", 128, 128, 11, 5, 33024, 256),
    ("L3 k3", 64, 64, 3, 1, 66048, 512), ("L3 k11d5", 64, 64, 11, 5, 66048, 512),
    ("L4 k3", 32, 32, 3, 1, 132096, 1024), ("L4 k7d3", 32, 32, 7, 3, 132096, 1024), ("L4 k11d5", 32, 32, 11, 5, 132096, 1024),
    ("est k3 T", 256, 256, 3, 1, 516, 2), ("est 1x1 T", 256, 1024, 1, 1, 516, 2), ("est ff2 T", 1024, 256, 1, 1, 516, 2),
    ("est k3 T/2", 256, 256, 3, 1, 258, 1), ("est k3 512", 512, 256, 3, 1, 516, 2), ("est qkv", 256, 384, 1, 1, 516, 2),
]
variants = [(0, -1), (1, -1), (2, -1), (4, -1), (7, -1)]
if len(sys.argv) > 1:
    variants = [tuple(int(x) for x in v.split(":")) for v in sys.argv[1:]]
print(f"B={B}; columns: dbg:cfg -> ms (TFLOP/s)")
only = os.environ.get("SHAPES")
for name, cin, cout, k, d, T, P in shapes:
    if only and not any(name.startswith(o) for o in only.split(",")):
        continue
    bb = B if not name.startswith("est") else int(os.environ.get("EST_B", "64"))
    flops = 2.0 * cin * cout * k * bb * T
    row = []
    for dbg, cfg in variants:
        ms = C.c_float()
        rc = fn(eng.h, cin, cout, k, d, bb, T, P, 5, dbg, cfg, C.byref(ms))
        row.append(f"{dbg}:{cfg} {ms.value:7.3f} ({flops / ms.value / 1e9:6.1f})" if rc == 0 else f"{dbg}:{cfg} ERR")
    print(f"{name:12s} " + " | ".join(row), flush=True)
