#!/bin/bash
# round 6, GPU call 24: the add / sub forms of the spectral automorphism with the body operand phi(body) +- a0 as 16-bit copies (HEAD) vs the i64 pre-pass
# (POULPY_DBG_AUTO_BODY16=0, experiment build)
OUT=gpurun_out/r6_run24; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "automorphism or trace or circuit or pack" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
line() { python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 2 --timing-steps 10 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print('%-8s %-58s %9.0f %-20s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], (d.get('parity_sample') or {}).get('ok'), r.get('kernel_ms')))"; }
{
for rep in 1 2 3; do
  for v in i64 t16; do
    unset POULPY_DBG_AUTO_BODY16
    [ $v = i64 ] && export POULPY_DBG_AUTO_BODY16=0
    line $v "--op automorphism_add"
    line $v "--op automorphism_add --limbs 16 --batch 512 --steps 20"
    line $v "--op automorphism_add --galois 1979"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
