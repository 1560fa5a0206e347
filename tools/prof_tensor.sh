#!/bin/bash
# rocprofv3 evidence for the GLWE tensoring (BASELINE configs[4], multiply half): kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes) and SQ counters of tools/bench_tensor.py.  Run on the GPU box through gpurun; summaries under gpurun_out/prof_tensor.
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_tensor; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --parity-samples 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_tensor.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/bench_tensor.py --steps 2 --warmup 1 --parity-samples 0 $@ > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/bench_tensor.py --steps 2 --warmup 1 --parity-samples 0 $@ > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq_a -- python3 $REPO/tools/bench_tensor.py --steps 2 --warmup 1 --parity-samples 0 $@ > $OUT/sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/sq_b -- python3 $REPO/tools/bench_tensor.py --steps 2 --warmup 1 --parity-samples 0 $@ > $OUT/sq_b.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob, collections, os
for kind in ("pmc_fetch", "pmc_write", "sq_a", "sq_b"):
    files = glob.glob(f"{kind}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:90], row.get("Counter_Name"))
            agg[k][0] += 1
            agg[k][1] += float(row.get("Counter_Value", 0))
    with open(f"{kind}_summary.txt", "w") as o:
        for (k, c), (n, v) in sorted(agg.items()):
            if "pz::" in k:
                o.write(f"{k}\t{c}\tdispatches={n}\tper_dispatch={v/max(n,1):.5g}\n")
    for f in files:
        os.remove(f)
files = sorted(glob.glob("trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
if files:
    with open("kernel_stats.txt", "w") as o:
        for r in csv.DictReader(open(files[-1])):
            if "pz::" in r.get("Name", ""):
                o.write("%s\tcalls=%s\tavg_ns=%s\ttotal_ns=%s\tpct=%s\n" % (r["Name"][:100], r.get("Calls"), r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
    print(open("kernel_stats.txt").read())
PY
find . -name "*kernel_trace.csv" -size +1M -delete
tail -2 trace.log
