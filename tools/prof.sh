#!/bin/bash
# rocprofv3 recipe for the bench (run on the GPU box through gpurun); outputs under gpurun_out/prof
set -x
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --parity-samples 0 --sustained-seconds 0"
ARGS_PMC="--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --parity-samples 0 --sustained-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS_PMC > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS_PMC > $OUT/pmc_write.log 2>&1
cd $OUT && find . -name "*.csv" | head -30
# keep only small summaries: stats + aggregated counters
python3 - <<'PY'
import csv, glob, collections, os
for kind in ("pmc_fetch", "pmc_write"):
    files = glob.glob(f"{kind}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:80], row.get("Counter_Name"))
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0))
    with open(f"{kind}_summary.txt", "w") as o:
        for (k, c), (n, v) in sorted(agg.items()):
            o.write(f"{k}\t{c}\tdispatches={n}\tsum={v:.1f}\tper_dispatch={v/max(n,1):.1f}\n")
    for f in files:
        os.remove(f)
PY
# steady-state averages.  With these ARGS (no parity sample, no per-kernel timing pass, no set-up calls: the placement tuner and its extra
# bench leg are gone since round 4) every pipeline kernel is dispatched warmup + steps = 2 + 20 times and the LAST 20 are exactly bench.py's
# timed region (round 3's "last 40" window mixed 19 timed launches with 21 of the placement leg).  The --stats average above also
# contains the warm-up launches (cold clocks: the maximum of the column).
python3 - <<'PY'
import csv, glob, collections, os
files = sorted(glob.glob("trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
if files:
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(files[-1])):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    with open("trace_steady.txt", "w") as o:
        o.write("# kernel, dispatches, avg ms over all, avg ms over the last 20 (= bench.py's timed region at --steps 20: warm-up 2 + timed 20, nothing behind it)\n")
        for k, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d in kv[1])):
            if "pz::" not in k: continue
            v.sort()
            d = [x[1] for x in v]
            last = d[-20:]
            o.write("%s\t%d\t%.4f\t%.4f\n" % (k[:70], len(d), sum(d) / len(d) / 1e6, sum(last) / len(last) / 1e6))
    print(open("trace_steady.txt").read())
PY
find . -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT $OUT/*
