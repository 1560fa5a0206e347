#!/usr/bin/env python3
"""Instruction mix of the loops of every kernel in an ISA dump: floating-point VALU, other VALU, scalar ALU, branches, memory.
Flags loops whose useful (floating-point) work sits among many branches / scalar instructions - the pattern behind the blind-rotation
fixes of round 3 (run-time tests inside unrolled loops, wrap arithmetic per row).

usage: isa_loop_mix.py file.s [name filter] [--all]
"""
import re, sys, subprocess
path = sys.argv[1]
args = [a for a in sys.argv[2:] if not a.startswith("--")]
flt = args[0] if args else ""
show_all = "--all" in sys.argv
txt = open(path).read().split("\n")
starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
names = [txt[i].split(":")[0] for i in starts]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for si, (st, nm) in enumerate(zip(starts, names)):
    d = dem[si].replace("void pz::", "").split("(")[0]
    if flt not in d: continue
    try: end = next(i for i in range(st, len(txt)) if "s_endpgm" in txt[i])
    except StopIteration: continue
    L = txt[st:end]
    labels = {}
    for i, l in enumerate(L):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    seen = set()
    for i, l in enumerate(L):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if not (m and m.group(1) in labels and labels[m.group(1)] < i): continue
        a = labels[m.group(1)]
        if (a, i) in seen: continue
        seen.add((a, i))
        seg = L[a:i]
        fp = sum(1 for x in seg if re.search(r"\bv_(fma|fmac|mul|add|pk_\w+)_f(64|32)", x))
        valu = sum(1 for x in seg if re.match(r"\s*v_", x))
        salu = sum(1 for x in seg if re.match(r"\s*s_", x) and "s_waitcnt" not in x and "s_nop" not in x and "s_cbranch" not in x and "s_branch" not in x)
        br = sum(1 for x in seg if "s_cbranch" in x or "s_branch" in x)
        vm = sum(1 for x in seg if re.match(r"\s*(global|buffer|flat)_(load|store)", x))
        ds = sum(1 for x in seg if re.match(r"\s*ds_", x))
        if fp < 32 or len(seg) > 6000: continue
        flag = br >= 6 or salu > 0.35 * fp or (valu - fp) > 0.5 * fp
        if flag or show_all:
            print(f"{d[:58]:58s} {m.group(1):10s} fp {fp:4d} valu-other {valu - fp:4d} salu {salu:4d} branches {br:3d} mem {vm:3d} lds {ds:3d}{'   <--' if flag else ''}")
