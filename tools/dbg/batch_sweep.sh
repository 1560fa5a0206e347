#!/bin/bash
# ciphertexts per device call vs rate, one shape (round 3: does keeping the intermediates under the 256 MiB Infinity Cache pay at small N?)
# usage: tools/dbg/batch_sweep.sh "<bench.py shape args>" b1 b2 ...   (env knobs pass through)
args="$1"; shift
for b in "$@"; do
  python bench.py $args --batch $b --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('batch', sys.argv[1], round(d['value']), '/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel_ms'])" $b
done
