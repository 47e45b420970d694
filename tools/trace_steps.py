#!/usr/bin/env python3
"""Per-pass totals of the conv kernels (conv_gemm_kernel + resblock_pair_kernel) from a rocprofv3 --kernel-trace CSV of
`bench.py`.  Every pass over the path issues the same number of conv launches (480 at config 2).  The LAST four passes of a
default run are un-overlapped, in this order: the HIP-event roofline pass, the batch-latency pass, the PCIe-inclusive pass and
the B=8 parity pass; the warm-up and timed passes before them are pipelined on two streams (their kernels overlap, so their
durations are not comparable with the roofline pass and are reported only as a group).
    python tools/trace_steps.py <rocprof output dir> [conv launches per pass: 480 at config 2]"""
import csv, glob, sys
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 480
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Dispatch_Id"]))
conv = [r for r in rows if "conv_gemm" in r["Kernel_Name"] or "resblock_pair" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
npass = len(conv) // n
print(f"{len(conv)} conv dispatches = {len(conv) / n:.2f} passes of {n}")
names = {npass - 4: "HIP-event roofline pass (un-overlapped)", npass - 3: "batch-latency pass (un-overlapped)",
         npass - 2: "PCIe-inclusive pass (un-overlapped)", npass - 1: "B=8 parity pass"}
head = conv[:(npass - 4) * n]
if head:
    t0 = min(int(r["Start_Timestamp"]) for r in head); t1 = max(int(r["End_Timestamp"]) for r in head)
    print(f"passes 0..{npass - 5} (warm-up + timed, two streams): {len(head)} conv launches, summed kernel time {sum(dur(r) for r in head) / 1e3:8.2f} ms, "
          f"span {(t1 - t0) / 1e6:8.2f} ms = {(t1 - t0) / 1e6 / (npass - 4):.2f} ms per pass")
for s in range(max(npass - 4, 0), npass):
    seg = conv[s * n:(s + 1) * n]
    tot = sum(dur(r) for r in seg)
    print(f"pass {s}: {len(seg)} conv launches, total {tot / 1e3:8.2f} ms, average {tot / len(seg):7.1f} us per launch   [{names.get(s, '')}]")
