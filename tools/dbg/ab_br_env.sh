#!/bin/bash
# same-box A/B of an environment knob over blind-rotation shapes: tools/dbg/ab_br_env.sh KNOB "v1 v2" "shapes" [cbt]
KNOB=$1; VALS=$2; SHAPES=$3; CBT=$4
for rep in 1 2 3; do
for v in $VALS; do
  export $KNOB=$v
  for sh in $SHAPES; do
    extra=""; [ "$sh" = big ] && extra="--batch ${BIG_BATCH:-256}"
    python tools/bench_blind_rotation.py --shape $sh $extra --cpu-cts 1 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%s=%s %-8s %9.0f rotations/s  parity %s' % ('$KNOB', '$v', '$sh', d['value'], d.get('parity_on_cpu_sample')))"
  done
  [ -z "$CBT" ] || python tools/bench_circuit_bootstrapping.py --batch 512 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%s=%s %-8s %9.0f bootstrappings/s parity %s' % ('$KNOB', '$v', 'circuit', d['value'], d.get('parity_on_cpu_sample')))"
done
done
