#!/bin/bash
# round 4, item 3 (second half): phi(body) gathered by the tail (default since round 4) vs prepared by a pre-pass (POULPY_DBG_AUTO_FOLD=0),
# same box, alternating; Galois elements 5 (rotation by one slot), -1 (conjugation), 5^7 and 3 (no locality between neighbouring
# coefficients); parity first
echo "== parity (automorphism / trace / config tests)"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "automorphism or trace or config5 or rotate" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --parity-samples 2"
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %9.0f /s  parity=%s  %s' % ('$1', d['value'], (d.get('parity_sample') or {}).get('ok'), {k: round(v,3) for k,v in d['roofline'].get('kernel_ms',{}).items()}))"; }
for op in automorphism automorphism_add; do
  for g in 5 -1 78125 3; do
    for rep in 1 2; do
      for fold in 1 0; do
        POULPY_DBG_AUTO_FOLD=$fold $B --op $op --galois $g 2>/dev/null | show "$op g=$g fold=$fold"
      done
    done
  done
done
echo "== 16 limbs (configs[4] rotate)"
for fold in 1 0 1 0; do POULPY_DBG_AUTO_FOLD=$fold $B --op automorphism --limbs 16 --batch 512 2>/dev/null | show "automorphism 16 limbs fold=$fold"; done
echo "== trace"
for fold in 1 0; do POULPY_DBG_AUTO_FOLD=$fold $B --op trace --steps 5 2>/dev/null | show "trace fold=$fold"; done
