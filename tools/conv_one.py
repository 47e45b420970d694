#!/usr/bin/env python3
"""Run one conv shape a few times (for rocprofv3 --pmc): python tools/conv_one.py Cin Cout K dil B T P [dbg] [cfg]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emojivoice_amd._lib import Engine  # noqa: E402

a = [int(x) for x in sys.argv[1:]]
cin, cout, k, d, B, T, P = a[:7]
dbg = a[7] if len(a) > 7 else 0
cfg = a[8] if len(a) > 8 else -1
eng = Engine(0)
fn = eng.lib.ev_dbg_conv_bench
fn.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_float)]
ms = C.c_float()
rc = fn(eng.h, cin, cout, k, d, B, T, P, 3, dbg, cfg, C.byref(ms))
print(rc, ms.value, 2.0 * cin * cout * k * B * T / ms.value / 1e9, "TFLOP/s")
