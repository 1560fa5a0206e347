#!/bin/bash
# timing ablation of the one-kernel rotation's phases (experiment build; results are WRONG with a mask set): POULPY_DBG_BR_SKIP bits
# 1 DFT passes, 2 product, 4 carry phase, 8 pack.  tools/dbg/br_phase_ablation.sh "ref cbt" 512
SHAPES="${1:-ref}"; B="${2:-512}"
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
for sh in $SHAPES; do for mask in 0 1 2 4 8 3 7 15; do
  POULPY_DBG_BR_SKIP=$mask POULPY_DBG_BR_FORM=${FORM:-2} python tools/bench_blind_rotation.py --shape $sh --batch $B --cpu-cts 0 --reps 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-6s batch %5d skip %2d  %7.3f ms  %s' % ('$sh', $B, $mask, d['ms_per_batch'], d.get('dispatch','')[:60]))"
done; done
