#!/bin/bash
# the SMI sampler guard: env of the box, default bench (sampler must run), bench under rocprofv3 --pmc (sampler must not run, no refused exec)
OUT=gpurun_out/r6_run34; mkdir -p $OUT
env | grep -i -E "preload|rocprof|rocp_" | cut -c1-200 > $OUT/env.txt; cat $OUT/env.txt
rm -f gpurun_out/.graft_exec_refused
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
grep "^{" $OUT/bench_default.json | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['sustained'])"
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --parity-samples 0 --sustained-seconds 1 > $REPO/$OUT/pmc.log 2>&1
cd $REPO; grep "^{" $OUT/pmc.log | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('under rocprofv3 --pmc:', d['value'], d['sustained'])"
find $OUT/pmc -name "*.csv" -delete
ls gpurun_out/.graft_exec_refused 2>/dev/null && wc -l gpurun_out/.graft_exec_refused || echo "no refused exec"
