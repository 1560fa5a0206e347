#!/bin/bash
# SQ counter pass for the bench kernels (run on the GPU box through gpurun)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --parity-samples 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/a -- python3 $REPO/bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/b -- python3 $REPO/bench.py $ARGS > $OUT/b.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections, os
for kind in ("a", "b"):
    files = glob.glob(f"{kind}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name"))
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0))
    with open(f"{kind}_summary.txt", "w") as o:
        for (k, c), (n, v) in sorted(agg.items()):
            if k.startswith("void pz") or k.startswith("pz::"):
                o.write(f"{k}\t{c}\tdispatches={n}\tper_dispatch={v/max(n,1):.4g}\n")
    for f in files: os.remove(f)
PY
tail -2 a.log
