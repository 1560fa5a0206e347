#!/bin/bash
# timeline of the duplex host path (rocprofv3 kernel + memory-copy trace of tools/bench_host_path.py --pinned): the last 16-ciphertext call
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_host; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $REPO/tools/bench_host_path.py --pinned > $OUT/run.log 2>&1
cd $OUT
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?")), r.get("Size", "")))
for f in glob.glob("**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kern " + r["Kernel_Name"][:40], ""))
ev.sort()
# the last 64 events before the end: print relative microseconds
tail = ev[-72:]
t0 = tail[0][0]
for s, e, name, size in tail:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name} {size}")
PY
