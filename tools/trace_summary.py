#!/usr/bin/env python3
"""Group a rocprofv3 kernel-trace CSV by (kernel, grid) with total/avg times; skip the first N dispatches (warm-up)."""
import collections, csv, glob, sys
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Dispatch_Id"]))
half = len(rows) // 2 if len(sys.argv) < 3 else int(sys.argv[2])
agg = collections.OrderedDict()
for r in rows[half:]:
    k = (r["Kernel_Name"].split("(")[0][-38:], int(r["Grid_Size_X"]) // 256)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in agg.items():
    print(f"{k[0]:40s} wgs={k[1]:7d} n={v[0]:4d} total={v[1] / 1e3:9.2f} ms avg={v[1] / v[0]:9.1f} us  {100 * v[1] / tot:5.1f}%")
print(f"total {tot / 1e3:.2f} ms")
