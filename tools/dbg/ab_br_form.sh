#!/bin/bash
# one-kernel blind rotation: ciphertexts per workgroup / workgroup size against the batch size (experiment build: POULPY_DBG_BR_FORM
# 2 = two ciphertexts per 512-thread workgroup, 1 = one, 3 = one per 256-thread workgroup);  tools/dbg/ab_br_form.sh "ref cbt" "64 128 256 512 1024"
SHAPES="${1:-ref}"; BATCHES="${2:-64 128 256 512 1024 2048}"
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
for sh in $SHAPES; do for b in $BATCHES; do for rep in 1 2; do for f in ${FORMS:-2 1 3}; do
  [ "$sh" != ref ] && [ "$f" -ge 3 ] && continue
  POULPY_DBG_BR_FORM=$f python tools/bench_blind_rotation.py --shape $sh --batch $b --cpu-cts 0 --reps 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-6s batch %5d form $f %9.0f rotations/s  %7.3f ms  %s' % ('$sh', $b, d['value'], d['ms_per_batch'], d.get('dispatch','')[:60]))"
done; done; done; done
