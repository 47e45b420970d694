#!/usr/bin/env python3
"""HBM traffic of the conv kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench command.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced
reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Counter unit: KiB."""
import collections, csv, glob, json, sys
def load(d, name):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == name:
            k = "conv" if ("conv_gemm" in r["Kernel_Name"] or "conv_split" in r["Kernel_Name"] or "conv_h16" in r["Kernel_Name"] or "resblock_pair" in r["Kernel_Name"] or "ln_mlp" in r["Kernel_Name"] or "ln_qkv" in r["Kernel_Name"] or "attn_out" in r["Kernel_Name"]) else "other"
            out[k] += float(r["Counter_Value"]); n[k] += 1
    return out, n
f, nf = load(sys.argv[1], "FETCH_SIZE")
w, nw = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
res = {"conv_launches": nf["conv"], "fetch_GB_raw": f["conv"] * 1024 / 1e9, "fetch_GB_corrected": 2 * f["conv"] * 1024 / 1e9,
       "write_GB": w["conv"] * 1024 / 1e9, "steps_profiled": steps}
res["hbm_GB_per_step"] = (res["fetch_GB_corrected"] + res["write_GB"]) / steps
res["hbm_MB_per_launch"] = 1e3 * (res["fetch_GB_corrected"] + res["write_GB"]) / max(nf["conv"], 1)
res["other_kernels_GB"] = (2 * f["other"] + w["other"]) * 1024 / 1e9
# which library the passes ran on: sha256 of the three sources the library is built from (bench.py compares it with the running library's)
import hashlib, os
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hsh = hashlib.sha256()
for rel in ("emojivoice_amd/csrc/ev_kernels.h", "emojivoice_amd/csrc/ev_engine.hip", "include/emojivoice.h"):
    hsh.update(open(os.path.join(here, rel), "rb").read())
res["library_source_sha16"] = hsh.hexdigest()[:16]
print(json.dumps(res))
