#!/bin/bash
# A/B library builds on the same box: `tools/dbg/ab_libs.sh [--ops "extra bench args"] lib1.so lib2.so ...` (paths relative to poulpy_amd/)
# prints products/s and the per-class kernel times of bench.py for each, two processes per library.
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --parity-samples 2"
if [ "$1" = "--args" ]; then ARGS="$ARGS $2"; shift 2; fi
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for i in 1 2; do
    python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-36s %9.0f /s  parity=%s  %s' % ('$lib', d['value'], (d.get('parity_sample') or {}).get('ok'), {k: round(v,3) for k,v in d['roofline'].get('kernel_ms',{}).items()}))"
  done
done
