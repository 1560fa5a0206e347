#!/bin/bash
# round 6, GPU call 26: glwe_trace with the 16-bit body operand + the shifted store on the f64 chain (k_inv_tail<..,RSH,7,SGN>): tests, then the trace lines
OUT=gpurun_out/r6_run26; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -k "trace or automorphism or circuit or pack" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -2
B="python bench.py --no-cpu-baseline --parity-samples 4 --sustained-seconds 0"
{
for rep in 1 2; do
  $B --op trace --steps 5 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%9.0f %s parity=%s %s' % (d['value'], d['unit'], d['parity_sample']['ok'], d['roofline']['kernel_ms']))"
  $B --op trace --steps 5 --limbs 16 --batch 512 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%9.0f %s parity=%s %s (16 limbs)' % (d['value'], d['unit'], d['parity_sample']['ok'], d['roofline']['kernel_ms']))"
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
