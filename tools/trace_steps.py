#!/usr/bin/env python3
"""Per-step totals of the conv kernels (conv_gemm_kernel + resblock_pair_kernel) from a rocprofv3 --kernel-trace CSV of
`bench.py`: every pass over the path issues the same number of conv launches, so consecutive groups of that many
dispatches are the steps (warm-up, timed steps, the HIP-event profile step, the B=8 parity pass).
    python tools/trace_steps.py <rocprof output dir> [launches per step, default 539]"""
import csv, glob, sys
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 539
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Dispatch_Id"]))
conv = [r for r in rows if "conv_gemm" in r["Kernel_Name"] or "resblock_pair" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{len(conv)} conv dispatches = {len(conv) / n:.2f} passes of {n}")
for s in range(len(conv) // n):
    seg = conv[s * n:(s + 1) * n]
    tot = sum(dur(r) for r in seg)
    print(f"pass {s}: {len(seg)} conv launches, total {tot / 1e3:8.2f} ms, average {tot / len(seg):7.1f} us per launch")
