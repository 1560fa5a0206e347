#!/usr/bin/env python3
"""Turns the output of tools/prof.sh (gpurun_out/prof) into the committed summaries profiles/rNN_kernel_stats.csv and
profiles/rNN_traffic.json.  HBM traffic per dispatch = 2 x FETCH_SIZE + WRITE_SIZE (separate PMC passes; on gfx950 FETCH_SIZE reports
half the bytes of wide streaming reads, MI355X_MICROARCH.md § HBM), in bytes (the counters are in KiB).
usage: python tools/prof_summary.py r02 [batch_per_launch]"""
import csv, glob, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prof = os.path.join(root, "gpurun_out", "prof")
stats = sorted(glob.glob(os.path.join(prof, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # newest last
if stats:
    rows = [r for r in csv.reader(open(stats[-1]))]
    with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w") as o:
        o.write(f"# {tag} — rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --parity-samples 0`\n")
        o.write(f"# (tools/prof.sh on the MI355X box; {batch} ciphertexts per launch of each pipeline kernel; default plan at N = 2^16: m1 = 256, m2 = 128)\n")
        w = csv.writer(o)
        for r in rows:
            if r and (r[0] == "Name" or r[0].startswith("pz::") or r[0].startswith("void pz::")):
                w.writerow(r)
def per_dispatch(kind):
    out = {}
    f = os.path.join(prof, f"{kind}_summary.txt")
    if not os.path.exists(f):
        return out
    for line in open(f):
        parts = line.rstrip("\n").split("\t")
        if len(parts) < 5:
            continue
        name = re.sub(r"^void ", "", parts[0]).split("(")[0]
        out[name] = float(parts[4].split("=")[1])
    return out
fetch, write = per_dispatch("pmc_fetch"), per_dispatch("pmc_write")
kern = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("pz::"):
        continue
    f, w_ = fetch.get(k, 0.0), write.get(k, 0.0)
    kern[k] = {"FETCH_SIZE_KB_per_dispatch": round(f, 1), "WRITE_SIZE_KB_per_dispatch": round(w_, 1),
               "hbm_bytes_per_dispatch_corrected": round((2 * f + w_) * 1024, 1)}
if kern:
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof.sh), bench default batch, default plan 256 x 128; "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md § HBM before summing", "batch_per_launch": batch, "kernels": kern},
              open(os.path.join(root, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
print("kernel stats:", bool(stats), "traffic kernels:", len(kern))
