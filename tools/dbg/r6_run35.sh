#!/bin/bash
# N = 2048 key switch / 3-limb shapes on k_small_one (variant -DPZ_SMALL_ONE_ALL=1) vs the two-kernel pipeline (HEAD's dispatch rule), after the both-columns form
OUT=gpurun_out/r6_run35; mkdir -p $OUT
line() { python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 2 --steps 100 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%-6s %-52s %10.0f %-20s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], d['parity_sample']['ok'], (d.get('roofline') or {}).get('kernel_ms')))"; }
{
for rep in 1 2; do for v in head soall; do
  if [ $v = head ]; then unset POULPY_HIP_LIB; else export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_$v.so; fi
  line $v "--n 2048 --limbs 4 --base2k 17 --op keyswitch"
  line $v "--n 2048 --limbs 3 --base2k 17 --op keyswitch"
  line $v "--n 2048 --limbs 2 --base2k 17"
  line $v "--n 2048 --limbs 4 --base2k 17 --op keyswitch --batch 4096"
done; done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-220
