#!/bin/bash
# round 6, GPU call 23: the 16-bit body pre-pass through LDS (k_automorphism_t16, one source read in whole lines) vs the gather kernels with a 16-bit store
# (POULPY_DBG_AUTO_T16_LDS=0) vs the i64 pre-pass + operand variant (POULPY_DBG_AUTO_BODY16=0); experiment build for the knobs
OUT=gpurun_out/r6_run23; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "automorphism or trace or circuit" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
line() { python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 2 --timing-steps 10 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print('%-14s %-58s %9.0f %-18s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], (d.get('parity_sample') or {}).get('ok'), r.get('kernel_ms')))"; }
{
for rep in 1 2 3; do
  for v in "i64" "t16-gather" "t16-lds"; do
    unset POULPY_DBG_AUTO_BODY16 POULPY_DBG_AUTO_T16_LDS
    [ $v = i64 ] && export POULPY_DBG_AUTO_BODY16=0
    [ $v = t16-gather ] && export POULPY_DBG_AUTO_T16_LDS=0
    line $v "--op automorphism --limbs 16 --batch 512 --steps 20"
    line $v "--op automorphism"
    line $v "--op automorphism --galois 1979 --limbs 16 --batch 512 --steps 20"
    line $v "--op automorphism --n 16384 --batch 4096 --steps 20"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
