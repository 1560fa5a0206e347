#!/bin/bash
# round 4: k_automorphism_chunk (4 outputs per thread, sources read in runs) for Galois elements without locality vs the plain gather
# (POULPY_DBG_AUTO_CHUNK=0), and forced for every element (=2); parity first (both forms)
for k in 1 2; do
echo "== parity, POULPY_DBG_AUTO_CHUNK=$k"
POULPY_DBG_AUTO_CHUNK=$k timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "automorphism or trace or config5 or rotate or pack or circuit" 2>&1 | tail -1
done
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --parity-samples 2"
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %9.0f /s  parity=%s  %s' % ('$1', d['value'], (d.get('parity_sample') or {}).get('ok'), {k: round(v,3) for k,v in d['roofline'].get('kernel_ms',{}).items()}))"; }
for op in automorphism automorphism_add; do
  for g in 5 -1 78125 3 25; do
    for k in 0 1 2; do
      POULPY_DBG_AUTO_CHUNK=$k $B --op $op --galois $g 2>/dev/null | show "$op g=$g chunk=$k"
    done
  done
done
echo "== trace"
for k in 0 1 0 1; do POULPY_DBG_AUTO_CHUNK=$k $B --op trace --steps 5 2>/dev/null | show "trace chunk=$k"; done
echo "== trace N=4096 (small-ring path: no pre-pass)"
for k in 0 1; do POULPY_DBG_AUTO_CHUNK=$k $B --n 4096 --limbs 4 --base2k 17 --op trace --steps 20 2>/dev/null | show "trace n4096 chunk=$k"; done
