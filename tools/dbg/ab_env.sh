#!/bin/bash
# A/B of environment knobs on one library and one box: tools/dbg/ab_env.sh [--args "bench args"] "VAR=a" "VAR=b" ...
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --parity-samples 2"
if [ "$1" = "--args" ]; then ARGS="$ARGS $2"; shift 2; fi
for kv in "$@"; do
  for i in 1 2; do
    env $kv python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-36s %9.0f /s  parity=%s  %s' % ('$kv', d['value'], d.get('parity_sample',{}).get('ok'), {k: round(v,3) for k,v in d['roofline'].get('kernel_ms',{}).items()}))"
  done
done
