import collections, csv, glob, sys, re
kt = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Dispatch_Id"]))
half = len(rows) // 2
agg = collections.OrderedDict()
first=int(rows[half]["Start_Timestamp"]); last=int(rows[-1]["End_Timestamp"])
for r in rows[half:]:
    k = re.sub(r"<.*", "", r["Kernel_Name"].split("(")[0])[-40:]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} n={v[0]:5d} total={v[1] / 1e3:9.2f} ms avg={v[1] / v[0]:9.1f} us  {100 * v[1] / tot:5.1f}%")
print(f"kernel total {tot / 1e3:.2f} ms; span {(last-first)/1e6:.2f} ms")
