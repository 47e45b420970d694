#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc + --kernel-trace CSV pair per kernel/grid (MFMA utilisation, wait shares, clock)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))}
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    x = disp[r["Dispatch_Id"]]
    x[r["Counter_Name"]] = float(r["Counter_Value"])
    x["name"], x["grid"] = r["Kernel_Name"], int(r["Grid_Size"])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = collections.OrderedDict()
for i in sorted(disp, key=int)[skip:]:
    x = disp[i]
    t = trace[i]
    dur = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
    key = (x["name"].split("(")[0][-40:], x["grid"] // 256, round(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1e7))
    a = agg.setdefault(key, collections.Counter())
    a["n"] += 1
    a["dur"] += dur
    for k, v in x.items():
        if isinstance(v, float):
            a[k] += v
print(f"{'kernel':42s} {'wgs':>7s} {'n':>4s} {'avg_us':>9s} {'mfma%':>6s} {'wait':>5s} {'winst':>5s} {'act':>5s} {'GHz':>5s}")
for (name, wgs, _), a in agg.items():
    n = a["n"]
    wc = max(a["SQ_WAVE_CYCLES"], 1)
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    print(f"{name:42s} {wgs:7d} {n:4d} {a['dur'] / n:9.1f} {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cyc * 1024, 1):6.1f} "
          f"{a['SQ_WAIT_ANY'] / wc:5.2f} {a['SQ_WAIT_INST_ANY'] / wc:5.2f} {a['SQ_ACTIVE_INST_ANY'] / wc:5.2f} {cyc / max(a['dur'], 1e-9) / 1e3:5.2f}")
