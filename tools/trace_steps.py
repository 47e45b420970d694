#!/usr/bin/env python3
"""Per-pass totals of the dominant-kernel family (conv_split_kernel / conv_split_bal_kernel / conv_gemm_kernel + resblock_pair_split_kernel + ln_mlp_kernel + attn_out_kernel) from a rocprofv3
--kernel-trace CSV of `bench.py --steps K --warmup W`.  Every pass over the path issues the same number of such launches
(420 at config 2 since round 2; 416 since the ResBlock chains of round 4: two chains replace six pair launches).  Order of the passes in a default run: W + K pipelined on two streams (their kernels overlap,
so their durations are reported only as a group), then 3 serial passes (`serial_ms_per_step`), the HIP-event roofline pass and
the stage-time pass, all un-overlapped on one stream; what follows (text-encoder timing, PCIe pass) is not segmented.
    python tools/trace_steps.py <kernel_trace.csv or rocprof output dir> <launches per pass> <W + K>"""
import csv, glob, sys
src = sys.argv[1]
kt = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 420
piped = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Dispatch_Id"]))
fam = ("conv_gemm", "conv_split", "conv_h16", "resblock_pair", "resblock_chain", "ln_mlp", "ln_qkv", "attn_out")
conv = [r for r in rows if any(f in r["Kernel_Name"] for f in fam)]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{len(conv)} dispatches of the conv family; {n} per pass")
head = conv[:piped * n]
t0 = min(int(r["Start_Timestamp"]) for r in head); t1 = max(int(r["End_Timestamp"]) for r in head)
print(f"passes 0..{piped - 1} (warm-up + timed, two streams): {len(head)} launches, summed kernel time {sum(dur(r) for r in head) / 1e3:8.2f} ms, "
      f"span {(t1 - t0) / 1e6:8.2f} ms = {(t1 - t0) / 1e6 / piped:.2f} ms per pass (under the profiler the two streams serialise)")
names = ["serial pass 1", "serial pass 2", "serial pass 3", "HIP-event roofline pass", "stage-time pass"]
for i, nm in enumerate(names):
    seg = conv[(piped + i) * n:(piped + i + 1) * n]
    if len(seg) < n:
        break
    tot = sum(dur(r) for r in seg)
    print(f"pass {piped + i}: {len(seg)} launches, total {tot / 1e3:8.2f} ms, average {tot / len(seg):7.1f} us per launch   [{nm}, un-overlapped]")
