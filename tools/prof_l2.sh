#!/bin/bash
# L1 (TCP) / L2 (TCC) traffic counters for the bench kernels (run on the GPU box through gpurun)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_l2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/bench.py $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $OUT/c -- python3 $REPO/bench.py $ARGS > $OUT/c.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections, os
for kind in ("a", "b", "c"):
    files = glob.glob(f"{kind}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:40], row.get("Counter_Name"))
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0))
    with open(f"{kind}_summary.txt", "w") as o:
        for (k, c), (n, v) in sorted(agg.items()):
            if "pz" in k:
                o.write(f"{k}\t{c}\tdispatches={n}\tper_dispatch={v/max(n,1):.4g}\n")
    for f in files: os.remove(f)
PY
tail -3 a.log b.log c.log
