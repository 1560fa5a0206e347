#!/bin/bash
# round 6, GPU call 38: per-kernel times (rocprofv3 --kernel-trace --stats) of the 16-limb rotation, the add form, glwe_trace and the N = 2^14 blind rotation at HEAD
REPO=$PWD; OUT=$REPO/gpurun_out/r6_run38; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -- python3 $REPO/$2 > $OUT/$1.log 2>&1
  python3 - $OUT/$1 $OUT/$1.txt <<'PY'
import csv, glob, os, sys
files = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
with open(sys.argv[2], "w") as o:
    for r in csv.DictReader(open(files[-1])):
        if "pz::" in r.get("Name", ""):
            o.write("%s\tcalls=%s\tavg_ns=%s\tpct=%s\n" % (r["Name"][:96], r.get("Calls"), r.get("AverageNs"), r.get("Percentage")))
PY
  find $OUT/$1 -name "*.csv" -delete; echo "== $1"; head -8 $OUT/$1.txt | cut -c1-170; }
B="bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 0 --no-kernel-timing --no-margin --steps 10 --warmup 2"
prof auto16 "$B --op automorphism --limbs 16 --batch 512"
prof autoadd "$B --op automorphism_add"
prof trace "$B --op trace --steps 2 --warmup 1"
prof brbig "tools/bench_blind_rotation.py --shape big --batch 1024 --reps 2"
