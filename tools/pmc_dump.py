#!/usr/bin/env python3
"""Print every counter of the conv_gemm dispatches of a rocprofv3 --pmc run, averaged over dispatches."""
import collections, csv, glob, sys
cc = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if "conv_gemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[1:] if len(v) > 1 else v
    print(f"{k:32s} {sum(v)/len(v):16.4g}  (n={len(v)})")
