#!/bin/bash
# round 6, GPU call 30: glwe_trace lines after the bench fix (clones synchronized before pin_key): HEAD (16-bit body operand + shifted store on the f64 chain)
# vs the i64 scheme (experiment build, POULPY_DBG_AUTO_BODY16=0)
OUT=gpurun_out/r6_run30; mkdir -p $OUT
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
B="python bench.py --no-cpu-baseline --parity-samples 2 --sustained-seconds 0 --op trace --steps 3"
one() { $B $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-8s %-26s %9.0f traces/s parity=%s margin=%.2g %s' % ('$1', '$2', d['value'], d['parity_sample']['ok'], d.get('rounding_margin'), (d.get('roofline') or {}).get('kernel_ms')))"; }
{
for rep in 1 2; do
for v in i64 t16; do
  unset POULPY_DBG_AUTO_BODY16; [ $v = i64 ] && export POULPY_DBG_AUTO_BODY16=0
  one $v ""
  one $v "--limbs 16 --batch 512"
  one $v "--limbs 12 --batch 512"
done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-230
