#!/usr/bin/env python3
"""Flag kernels whose ISA waits for vector-memory loads one at a time.

usage: isa_serial_loads.py file.s [name filter]     (file.s from `hipcc -S --cuda-device-only ...`)

For every kernel: the number of global/buffer loads, and how many of them are followed within WINDOW instructions by an
`s_waitcnt vmcnt(0)` with no other load in between ("lone" loads: each pays a full memory latency by itself).  Round 3: this is how
the serialized inter-pass twiddle loads of k_small_inv (16 x vmcnt(0)) and the collapsed ping-pong of its product loop were found.
"""
import re, sys, subprocess

WINDOW = 6
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name = None
rows = []
cur = []
def flush():
    if name is None: return
    loads = lone = 0
    i = 0
    ins = [l for l in cur if l and not l.startswith((";", ".")) and not l.endswith(":")]
    for i, l in enumerate(ins):
        if re.match(r"(global|buffer|flat)_load", l):
            loads += 1
            for j in range(i + 1, min(i + 1 + WINDOW, len(ins))):
                if re.match(r"(global|buffer|flat)_load", ins[j]): break
                if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
                    lone += 1
                    break
    rows.append((name, loads, lone))
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush()
        name = m.group(1); cur = []
        continue
    if name is not None:
        cur.append(line.strip())
        if "s_endpgm" in line:
            flush(); name = None
names = [r[0] for r in rows]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
except Exception:
    dem = names
for (n, loads, lone), d in zip(rows, dem):
    d = d.replace("void pz::", "").split("(")[0]
    if flt in d and lone >= 3:
        print(f"{d:64s} loads {loads:4d}  lone {lone:4d}")
