#!/bin/bash
# round 4: plain spectral glwe_automorphism with its body-less columns on the sign-only variant of the f64 tail (default) vs every column on
# the operand variant (POULPY_DBG_AUTO_SGN=0); parity first
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "automorphism or trace or config5 or rotate or pack or circuit" 2>&1 | tail -1
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --parity-samples 2"
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %9.0f /s  parity=%s  %s' % ('$1', d['value'], (d.get('parity_sample') or {}).get('ok'), {k: round(v,3) for k,v in d['roofline'].get('kernel_ms',{}).items()}))"; }
for rep in 1 2; do for k in 0 1; do
  POULPY_DBG_AUTO_SGN=$k $B --op automorphism 2>/dev/null | show "automorphism g=5 sgn=$k"
  POULPY_DBG_AUTO_SGN=$k $B --op automorphism --galois 78125 2>/dev/null | show "automorphism g=5^7 sgn=$k"
  POULPY_DBG_AUTO_SGN=$k $B --op automorphism --limbs 16 --batch 512 2>/dev/null | show "automorphism 16 limbs sgn=$k"
  POULPY_DBG_AUTO_SGN=$k $B --op automorphism --n 8192 --steps 100 2>/dev/null | show "automorphism N=8192 sgn=$k"
done; done
