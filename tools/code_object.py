#!/usr/bin/env python3
"""Read the gfx950 code object embedded in libemojivoice_hip.so and list every kernel's register / scratch census.

    python tools/code_object.py [lib.so]        one line per kernel: vgpr agpr sgpr vgpr-spills sgpr-spills scratch bytes LDS

The library is a host ELF whose .hip_fatbin section holds a clang offload bundle; entry `hipv4-amdgcn-amd-amdhsa--gfx950` is the device
ELF, whose NT_AMDGPU_METADATA note (msgpack, printed as YAML by `llvm-readelf --notes`) carries per kernel `.vgpr_count`, `.agpr_count`,
`.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` (scratch bytes per lane).  tests/test_code_object.py asserts on it.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_elf(lib_path: str, arch: str = "gfx950") -> bytes:
    data = open(lib_path, "rb").read()
    at = data.find(MAGIC)
    if at < 0:
        raise RuntimeError(f"{lib_path}: no uncompressed clang offload bundle found")
    n = struct.unpack_from("<Q", data, at + len(MAGIC))[0]
    off = at + len(MAGIC) + 8
    for _ in range(n):
        o, sz, tl = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if triple.startswith("hip") and triple.rstrip("-").endswith(arch):
            return data[at + o: at + o + sz]
    raise RuntimeError(f"{lib_path}: no {arch} code object in the bundle")


def kernels(lib_path: str):
    """[{name, demangled, vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}]"""
    elf = device_elf(lib_path)
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    out = []
    for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        block = ".agpr_count:" + block
        rec = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                    "group_segment_fixed_size", "max_flat_workgroup_size"):
            m = re.search(r"\." + key + r":\s+(\d+)", block)
            rec[key] = int(m.group(1)) if m else -1
        m = re.search(r"\.name:\s+(\S+)", block)
        rec["name"] = m.group(1) if m else "?"
        out.append(rec)
    names = "\n".join(r["name"] for r in out)
    dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.split("\n")
    for r, d in zip(out, dem):
        r["demangled"] = d.replace("void ", "", 1)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "emojivoice_amd", "lib", "libemojivoice_hip.so")
    ks = kernels(lib)
    print(f"# {lib}: {len(ks)} kernels; vgpr agpr sgpr | vgpr-spills sgpr-spills scratch-bytes/lane | LDS")
    for r in sorted(ks, key=lambda r: (-r["private_segment_fixed_size"], r["demangled"])):
        print(f"{r['vgpr_count']:4d} {r['agpr_count']:4d} {r['sgpr_count']:4d} | {r['vgpr_spill_count']:4d} {r['sgpr_spill_count']:4d} {r['private_segment_fixed_size']:5d} | "
              f"{r['group_segment_fixed_size']:6d}  {r['demangled'][:120]}")
