"""Registers, scratch and LDS of every kernel in a built libtwgpu.so (from the code object's metadata; no GPU needed).

    python profiles/tools/kernel_resources.py [lib] [substring ...]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = data.find(magic)
    out = []
    while at >= 0:
        n = struct.unpack_from("<Q", data, at + len(magic))[0]
        p = at + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx" in triple and size:
                out.append((triple, data[at + off:at + off + size]))
        at = data.find(magic, at + 1)
    return out


def kernels(path):
    rows = []
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            blk = ".agpr_count:" + blk
            def g(key):
                m = re.search(r"\." + key + r":\s+('?)([^\n']+)\1", blk)
                return m.group(2).strip() if m else ""
            name = g("name")
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"^void tw::|^tw::", "", dem)
            dem = re.sub(r"\(tw::.*$|\(.*$", "", dem)
            rows.append(dict(kernel=dem, vgpr=int(g("vgpr_count") or 0), agpr=int(g("agpr_count") or 0), sgpr=int(g("sgpr_count") or 0),
                             scratch=int(g("private_segment_fixed_size") or 0), lds=int(g("group_segment_fixed_size") or 0),
                             vspill=int(g("vgpr_spill_count") or 0), sspill=int(g("sgpr_spill_count") or 0)))
    return sorted(rows, key=lambda r: r["kernel"])


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(REPO, "traceweaver_amd", "lib", "libtwgpu.so")
    pats = [a for a in sys.argv[1:] if not os.path.exists(a)]
    print("| kernel | VGPR | AGPR | SGPR | scratch B/lane | LDS B | VGPR spills | SGPR spills | waves/SIMD by registers |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in kernels(lib):
        if pats and not any(p in r["kernel"] for p in pats):
            continue
        tot = max(r["vgpr"] + r["agpr"], 1)   # unified file of 512 per SIMD lane, allocated in blocks of 8
        occ = min(8, 512 // ((tot + 7) // 8 * 8))
        print("| `%s` | %d | %d | %d | %d | %d | %d | %d | %d |" % (r["kernel"], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"], r["vspill"], r["sspill"], occ))
