#!/usr/bin/env python3
"""Assemble DESIGN.md (contract) and NOTES.md (lab notebook) from tools/docs/*.md, filling @@PLACEHOLDERS@@ from a bench line.

    python tools/docs/assemble.py <bench_default.json> [key=value ...]

The skeletons are the round-3 DESIGN.md cut at its section boundaries (sections 1-2 and 4-7 kept verbatim, the old section 3 and status
tables moved to NOTES.md); design_sec3.md / design_sec8.md / notes_round4.md are this round's text."""
import json
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
repo = os.path.dirname(os.path.dirname(here))
rd = lambda n: open(os.path.join(here, n)).read()
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r, c5, f32 = d["roofline"], d.get("config5") or {}, d.get("value_fp32_mfma") or {}
cpu = d.get("cpu_baseline") or {}
tr = r.get("traffic") or {}
vals = {
    "VALUE": f"{d['value']:.0f}", "XRT": f"{d['value']:.0f}", "MS": f"{d['ms_per_step']:.1f}", "SERIAL": f"{d['serial_ms_per_step']:.1f}",
    "CFM": f"{d['stage_ms']['cfm_decode']:.1f}", "VOC": f"{d['stage_ms']['hifigan']:.1f}", "CONVMS": f"{r['conv_ms_per_step']:.1f}",
    "FAMTF": f"{r['family_tflops']:.0f}", "CFMTF": f"{r['tflops_cfm_convs']:.0f}", "VOCTF": f"{r['tflops_hifigan_convs']:.0f}",
    "ACH": f"{r['achieved']:.0f}", "FRAC": f"{r['frac']:.2f}", "F32TF": f"{r['fp32_mfma_family']['achieved']:.0f}", "F32FRAC": f"{r['fp32_mfma_family']['frac']:.2f}",
    "HBMGB": f"{tr.get('GB_per_step', 0):.0f}", "NFAM": str(r.get("family_launches_per_step", "")), "NSPLIT": str(r.get("launches_per_step", "")),
    "NF32": str((r.get("fp32_mfma_family") or {}).get("launches_per_step", "")), "MELLINF": f"{cpu.get('parity_mel_linf', 0):.1e}", "WAVRMS": f"{cpu.get('parity_wav_rms', 0):.1e}",
    "FP32VALUE": f"{f32.get('value', 0):.0f}", "FP32MS": f"{f32.get('ms_per_step', 0):.1f}", "FP32MEL": f"{f32.get('mel_linf_vs_headline_arithmetic', 0):.1e}",
    "FP32WAV": f"{f32.get('wav_rms_vs_headline_arithmetic', 0):.1e}", "CPUVALUE": f"{cpu.get('value', 0):.1f}",
    "C5WARM": f"{(c5.get('warm') or {}).get('p50_ms', 0)}", "C5COLD": f"{(c5.get('cold_length') or {}).get('p50_ms', 0)}",
    "C5GRAPH": f"{(c5.get('warm_graph_replay') or {}).get('p50_ms', 0)}",
}
import re
m = re.search(r"workspace \(([0-9.]+) GB", (d.get("config") or {}).get("memory", ""))
if m:
    vals["WSGB"] = m.group(1)
    vals["TOTGB"] = f"{2 * (0.45 + float(m.group(1))):.0f}"
c4 = d.get("config4") or {}
if c4:
    vals["CONFIG4"] = "CFM decode ms at n = " + ", ".join(f"{e['ode_steps']}: {e['cfm_ms']}" for e in c4["sweep"]) + \
        "; mel MSE vs n = 50: " + ", ".join(f"{e['mel_mse_vs_50']:.1e}" for e in c4["sweep"]) + \
        "; mel L∞ vs the CPU oracle at the same n ≤ " + f"{max(e['mel_linf_vs_cpu_same_n'] for e in c4['sweep']):.1e}"
try:    # the PMC traffic file of this round, when the bench line quoted an older one
    tj = json.load(open(os.path.join(repo, "profiles", "r04_conv_hbm_traffic_pmc.json")))
    vals["HBMGB"] = f"{tj['hbm_GB_per_step']:.0f}"
except Exception:
    pass
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    vals[k] = v

def fill(t):
    for k, v in vals.items():
        t = t.replace("@@" + k + "@@", v)
    return t

design = rd("design_skeleton.md").replace("@@SEC3@@", rd("design_sec3.md")).replace("@@SEC8@@", rd("design_sec8.md"))
notes = rd("notes_skeleton.md").replace("@@ROUND4_NOTES@@", rd("notes_round4.md"))
open(os.path.join(repo, "DESIGN.md"), "w").write(fill(design))
open(os.path.join(repo, "NOTES.md"), "w").write(fill(notes))
left = [w for w in (fill(design) + fill(notes)).split("@@")[1::2] if w.isupper() or "_" in w]
print("unfilled placeholders:", sorted(set(left)))
