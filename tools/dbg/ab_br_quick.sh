#!/bin/bash
# quick same-box A/B of blind-rotation shapes over library builds: tools/dbg/ab_br_quick.sh "ref cbt" lib1.so lib2.so  (libs relative to poulpy_amd/; 3 alternating rounds)
SHAPES="$1"; shift
for rep in 1 2 3; do
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for sh in $SHAPES; do
    extra=""; [ "$sh" = big ] && extra="--batch ${BIG_BATCH:-256}"
    python tools/bench_blind_rotation.py --shape $sh $extra --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-40s %-8s %9.0f rotations/s  margin %.2e' % ('$lib', '$sh', d['value'], d.get('rounding_margin') or 0))"
  done
done
done
