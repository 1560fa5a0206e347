#!/bin/bash
for c in 4 8 16 32 64 128; do
  python bench.py --steps 3 --warmup 1 --chunk $c --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('chunk', $c, 'value %.0f' % d['value'], 'ms/step %.2f' % d['ms_per_step'], r['kernel_ms'])
"
done
