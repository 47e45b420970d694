"""Per-basic-block instruction histogram of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_blocks.py listing.s <kernel-name-substring> [min_instr]

Answers "where do the scratch_load / scratch_store sit: K loop or prologue / epilogue?" (VERDICT round 3, item 1).
"""
import re
import sys


def blocks_of(path, needle):
    txt = open(path).read()
    m = re.search(r"\n(_Z\w*" + re.escape(needle) + r"\w*):[^\n]*\n", txt)
    if not m:
        raise SystemExit(f"no kernel matching {needle}")
    body = txt[m.end():]
    body = body[: body.index(".end_amdhsa_kernel")] if ".end_amdhsa_kernel" in body else body
    cur, order, blocks = "entry", [], {}
    keys = ("v_mfma", "scratch_store", "scratch_load", "v_accvgpr_write", "v_accvgpr_read", "v_writelane", "v_readlane", "buffer_load",
            "ds_read", "ds_write", "s_waitcnt", "s_barrier", "v_cvt", "s_cbranch")
    for line in body.split("\n"):
        lab = re.match(r"(\.LBB\d+_\d+):", line)
        if lab:
            cur = lab.group(1)
        t = line.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        if cur not in blocks:
            blocks[cur] = dict(n=0, **{k: 0 for k in keys})
            order.append(cur)
        b = blocks[cur]
        b["n"] += 1
        for k in keys:
            if t.startswith(k):
                b[k] += 1
    return m.group(1), order, blocks


def main():
    path, needle = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    name, order, blocks = blocks_of(path, needle)
    tot = {}
    print(name)
    for k in order:
        b = blocks[k]
        for kk, v in b.items():
            tot[kk] = tot.get(kk, 0) + v
        if b["v_mfma"] or b["scratch_store"] or b["scratch_load"] or b["n"] >= min_n:
            print(f"  {k:12s} " + " ".join(f"{kk.replace('v_accvgpr_', 'acc_').replace('scratch_', 'scr_')}={v}" for kk, v in b.items() if v))
    print("  total        " + " ".join(f"{kk}={v}" for kk, v in tot.items() if v))


if __name__ == "__main__":
    main()
