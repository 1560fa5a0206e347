#!/usr/bin/env python3
"""Register / scratch table of every kernel INSIDE a built libpoulpy_hip.so, read from the code objects' metadata notes
(llvm-objdump --offloading + llvm-readelf --notes): what actually ships, not what a fresh compile would produce (tools/kres.py).
usage: python tools/kres_so.py [lib.so] [filter]        prints: name vgpr= scratch= sgpr_spill= lds="""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def demangle(names):
    if not names:
        return []
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", n).replace("void pz::", "").replace("pz::", "") for n in out]


def kernel_table(lib=None):
    """[{name, vgpr, scratch, vgpr_spill, sgpr_spill, lds, wg}] for every kernel of the library's gfx950 code objects."""
    lib = lib or os.path.join(ROOT, "poulpy_amd", "libpoulpy_hip.so")
    rows = []
    with tempfile.TemporaryDirectory(prefix="kres_so_") as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)   # llvm-objdump writes the bundles next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], capture_output=True, text=True, check=True)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            cur = None
            for line in notes.splitlines():
                if re.match(r"\s+- \.", line):
                    cur = {}
                    rows.append(cur)
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
                if m and cur is not None and m.group(1) in ("name", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                            "group_segment_fixed_size", "max_flat_workgroup_size"):
                    cur[m.group(1)] = m.group(2)
    rows = [r for r in rows if "name" in r and "vgpr_count" in r]
    names = demangle([r["name"] for r in rows])
    return [{"name": n, "vgpr": int(r["vgpr_count"]), "scratch": int(r.get("private_segment_fixed_size", 0)), "vgpr_spill": int(r.get("vgpr_spill_count", 0)),
             "sgpr_spill": int(r.get("sgpr_spill_count", 0)), "lds": int(r.get("group_segment_fixed_size", 0)), "wg": int(r.get("max_flat_workgroup_size", 0))}
            for n, r in zip(names, rows)]


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    lib = args[0] if args and args[0].endswith(".so") else None
    flt = args[-1] if args and not args[-1].endswith(".so") else ""
    for r in kernel_table(lib):
        if flt in r["name"]:
            print(f"{r['name']:72s} vgpr={r['vgpr']} scratch={r['scratch']} vgpr_spill={r['vgpr_spill']} sgpr_spill={r['sgpr_spill']} lds={r['lds']} wg={r['wg']}")
