#!/bin/bash
# round 6, GPU call 21: the automorphism family's body operand as 16-bit tile-order copies (POULPY_DBG_AUTO_BODY16=1, HEAD) vs the i64 pre-pass (=0)
OUT=gpurun_out/r6_run21; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "automorphism or trace or circuit" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
line() { python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 2 --timing-steps 10 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print('%-9s %-58s %9.0f %-18s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], (d.get('parity_sample') or {}).get('ok'), r.get('kernel_ms')))"; }
{
for rep in 1 2 3; do
  for v in 0 1; do
    export POULPY_DBG_AUTO_BODY16=$v
    line body16=$v "--op automorphism --limbs 16 --batch 512 --steps 20"
    line body16=$v "--op automorphism"
    line body16=$v "--op automorphism_add"
    line body16=$v "--op automorphism --galois 1979 --limbs 16 --batch 512 --steps 20"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
