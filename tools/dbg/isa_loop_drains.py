#!/usr/bin/env python3
"""List inner loops that issue several vector loads and drain to `s_waitcnt vmcnt(0)` inside the loop body (a software pipeline the
machine scheduler has merged back into load-all / wait-all / use-all).

usage: isa_loop_drains.py file.s [name filter]
"""
import re, sys, subprocess
path = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
kern = None; lines = []
out = []
def scan(name, body):
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]; seg = body[a:i]
            if any(re.match(r"^\.LBB", x) for x in seg[1:]): continue   # innermost only
            loads = sum(1 for x in seg if re.match(r"\s*(global|buffer|flat)_load", x))
            waits = [re.search(r"vmcnt\((\d+)\)", x).group(1) for x in seg if "s_waitcnt" in x and "vmcnt" in x]
            valu = sum(1 for x in seg if re.match(r"\s*v_", x))
            if loads >= 4 and "0" in waits:
                out.append((name, m.group(1), loads, valu, ",".join(waits)))
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m: kern = m.group(1); lines = []; continue
    if kern:
        lines.append(line.rstrip())
        if "s_endpgm" in line: scan(kern, lines); kern = None
names = sorted(set(o[0] for o in out))
dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")))
for n, lab, loads, valu, waits in out:
    d = dem[n].replace("void pz::", "").split("(")[0]
    if flt in d: print(f"{d:60s} {lab:12s} loads {loads:3d} valu {valu:4d} vmcnt waits [{waits}]")
