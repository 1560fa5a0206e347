#!/bin/bash
# round 6, GPU call 27: glwe_trace at 16 limbs reported parity=False with the 16-bit body operand - is it the new form? experiment build, BODY16 = 0 / 1, 8 and 16 limbs, 2 ciphertext sizes
OUT=gpurun_out/r6_run27; mkdir -p $OUT
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
B="python bench.py --no-cpu-baseline --parity-samples 4 --sustained-seconds 0 --no-kernel-timing"
{
for v in 0 1; do
  export POULPY_DBG_AUTO_BODY16=$v
  for args in "--limbs 16 --batch 512" "--limbs 16 --batch 64" "--limbs 12 --batch 64" "--limbs 9 --batch 64" "--limbs 8 --batch 64"; do
    $B --op trace --steps 2 $args 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('body16=$v %-24s %9.0f %s parity=%s margin=%s' % ('$args', d['value'], d['unit'], d['parity_sample'], d.get('rounding_margin')))"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-300
