#!/usr/bin/env python3
"""Per-kernel marginal time on an in-order stream from a rocprofv3 --kernel-trace CSV: for the last N dispatches, the gap from the
previous kernel's end to this kernel's end is attributed to this kernel (what the launch adds to the wall time), next to its own
duration.    python tools/trace_marginal.py <rocprof output dir> <last N dispatches> <rows to print>"""
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-int(sys.argv[2]):]
agg=collections.defaultdict(lambda:[0,0,0])
prev_end=int(rows[0]["End_Timestamp"])
for r in rows[1:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    k=r["Kernel_Name"][:60]+" g"+r["Grid_Size_X"]+"x"+r["Grid_Size_Y"]
    agg[k][0]+=e-prev_end; agg[k][1]+=1; agg[k][2]+=e-s
    prev_end=e
tot=sum(v[0] for v in agg.values())
print("span ms",tot/1e6, "launches", len(rows))
for k,v in sorted(agg.items(),key=lambda kv:-kv[1][0])[:int(sys.argv[3])]: print("%7.3f ms %4d  marginal %6.2f us  duration %6.2f us  %s"%(v[0]/1e6,v[1],v[0]/v[1]/1e3,v[2]/v[1]/1e3,k))
