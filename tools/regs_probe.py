#!/usr/bin/env python3
"""Register / scratch census of the kernels on the config-2 launch list (tools/regs_probe.hip), compiled alone: seconds, not minutes.

    python tools/regs_probe.py [extra hipcc flags]      one line per kernel: VGPR AGPR scratch vgpr-spills sgpr-spills occupancy
    REGS_OUT=/tmp/x.s python tools/regs_probe.py        keeps the assembly there (tools/isa_blocks.py reads it)
"""
import os
import re
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
out = os.environ.get("REGS_OUT", "/tmp/regs_probe.s")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S", "-o", out,
       os.path.join(here, "regs_probe.hip"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
r = subprocess.run(cmd, capture_output=True, text=True)
txt = r.stderr
if r.returncode:
    print(txt[-4000:])
    sys.exit(1)
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not any(k in dem for k in ("h16", "attn_out", "ln_mlp", "_bal_", "_sk_", "conv_gemm_kernel")):
        continue
    print("%-66s vgpr %3d agpr %3d scratch %4d vspill %3d sspill %3d occ %d" % (dem[5:70], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                                               g("VGPRs Spill"), g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
