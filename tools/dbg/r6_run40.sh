#!/bin/bash
# round 6, GPU call 40: add / sub forms and glwe_trace with the OTHER column's operand as 16-bit copies too (pass 1 leaves its input as a side copy: k_fwd_pass1_w16;
# HEAD) vs POULPY_DBG_AUTO_SIDE16=0 (experiment build); automorphism / trace / bootstrap tests first
OUT=gpurun_out/r6_run40; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -k "automorphism or trace or circuit or pack" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -2
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
line() { python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 2 --timing-steps 10 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print('%-7s %-52s %9.0f %-20s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], (d.get('parity_sample') or {}).get('ok'), r.get('kernel_ms')))"; }
{
for rep in 1 2 3; do
  for v in side0 side16; do
    unset POULPY_DBG_AUTO_SIDE16; [ $v = side0 ] && export POULPY_DBG_AUTO_SIDE16=0
    line $v "--op automorphism_add"
    line $v "--op automorphism_add --limbs 16 --batch 512 --steps 20"
    line $v "--op trace --steps 3"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
