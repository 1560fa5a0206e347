#!/bin/bash
# round 6, GPU call 19: per-kernel times of the one-call multiply + relinearize: 16-bit store addressing (HEAD: base + constant offsets; variant: from the index)
OUT=$PWD/gpurun_out/r6_run19; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for v in head addr0; do
  if [ $v = addr0 ]; then export POULPY_HIP_LIB=$REPO/poulpy_amd/variants/libpoulpy_hip_addr0.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $REPO/tools/bench_tensor.py --steps 5 --warmup 1 --parity-samples 0 --relin --one-call > $OUT/trace_$v.log 2>&1
  python3 - $OUT/trace_$v $OUT/kernel_stats_$v.txt <<'PY'
import csv, glob, os, sys
files = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
with open(sys.argv[2], "w") as o:
    for r in csv.DictReader(open(files[-1])):
        if "pz::" in r.get("Name", ""):
            o.write("%s\tcalls=%s\tavg_ns=%s\n" % (r["Name"][:86], r.get("Calls"), r.get("AverageNs")))
PY
  find $OUT/trace_$v -name "*kernel_trace.csv" -size +1M -delete
  echo "== $v"; cat $OUT/kernel_stats_$v.txt
  grep "^{" $OUT/trace_$v.log | tail -1 | cut -c1-300
done
cd $REPO
for rep in 1 2; do for v in head addr0; do
  if [ $v = addr0 ]; then export POULPY_HIP_LIB=$REPO/poulpy_amd/variants/libpoulpy_hip_addr0.so; else unset POULPY_HIP_LIB; fi
  python tools/bench_tensor.py --parity-samples 1 --relin --one-call 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-8s %8.0f %s parity=%s %s' % ('$v', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"
done; done
