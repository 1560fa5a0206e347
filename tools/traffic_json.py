#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the two PMC summaries of tools/prof.sh (FETCH_SIZE and WRITE_SIZE in separate passes, KB per dispatch;
FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM before summing): what bench.py's roofline.traffic reads.

    python tools/traffic_json.py gpurun_out/r05_pmc_fetch_summary.txt gpurun_out/r05_pmc_write_summary.txt profiles/r05_traffic.json [batch]"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def kernel_signatures(names, lib=None):
    """{kernel: "vgpr/scratch/sgpr_spill/lds/wg"} from the code-object metadata of the library the counters were taken on (tools/kres_so.py): bench.py
    quotes the committed counters only while the kernel it runs still carries the same signature."""
    try:
        import kres_so
        table = {("pz::" + r["name"]): r for r in kres_so.kernel_table(lib)}
    except Exception as e:   # no llvm tools: no signatures, and bench.py then reports no traffic
        return {"error": str(e)}
    return {k: "%d/%d/%d/%d/%d" % (table[k]["vgpr"], table[k]["scratch"], table[k]["sgpr_spill"], table[k]["lds"], table[k]["wg"]) for k in names if k in table}


def read(path):
    out = {}
    for ln in open(path):
        parts = ln.rstrip("\n").split("\t")
        if len(parts) < 5 or "pz::" not in parts[0]:
            continue
        name = re.sub(r"^void ", "", parts[0])
        name = re.sub(r"\(.*", "", name)
        out[name] = float(parts[4].split("=")[1])
    return out


def main():
    fetch, write = read(sys.argv[1]), read(sys.argv[2])
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KB_per_dispatch": f, "WRITE_SIZE_KB_per_dispatch": w, "hbm_bytes_per_dispatch_corrected": (2 * f + w) * 1024.0}
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof.sh), bench default batch, default plan 256 x 128; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md section HBM before summing", "batch_per_launch": batch, "kernels": kernels,
           "kernel_signatures": kernel_signatures(list(kernels))}
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k[:70]:70s} {v['hbm_bytes_per_dispatch_corrected'] / 1e9:8.3f} GB per dispatch")


if __name__ == "__main__":
    main()
