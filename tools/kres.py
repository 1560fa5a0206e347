#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the HIP library (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py [filter-substring]"""
import re, subprocess, sys, os
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "poulpy_amd", "csrc")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
# kernels live in the launch_*.hip translation units; `--tu launch_mid` restricts the (slow) compile to one of them
tus = sorted(f for f in os.listdir(root) if f.startswith("launch_") and f.endswith(".hip"))
if "--tu" in sys.argv:
    tus = [sys.argv[sys.argv.index("--tu") + 1] + ".hip"]
    flt = "" if flt == "--tu" else flt
out = ""
for tu in tus:
    out += subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c",
                           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/kres.o", tu],
                          cwd=root, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    if flt in r["name"]:
        n = re.sub(r"\(.*", "", r["name"]).replace("void pz::", "")
        print(f"{n:48s} vgpr={r.get('vgpr')} scratch={r.get('scratch')} occ={r.get('occ')} lds={r.get('lds')}")
