#!/bin/bash
# which factor makes bench.py --op trace --limbs 12 report parity False (margin 0.5)? batch, pinning, graphs
OUT=gpurun_out/r6_run29; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --parity-samples 2 --sustained-seconds 0 --no-kernel-timing --op trace --steps 2 --limbs 12"
one() { $B $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-44s %9.0f parity=%s margin=%s' % ('$1 $2', d['value'], d['parity_sample']['ok'], d.get('rounding_margin')))"; }
{
one "" "--batch 2"
one "" "--batch 8"
one "" "--batch 16"
one "" "--batch 64"
one "" "--batch 64 --no-pin-key"
POULPY_DBG_GRAPHS=0 one "graphs=0" "--batch 64"
one "" "--batch 64 --warmup 0 --no-margin"
one "" "--batch 64 --limbs 10"
one "" "--batch 64 --limbs 11"
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
