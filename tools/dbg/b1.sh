#!/bin/bash
for args in "--steps 5 --warmup 2" "--steps 5 --warmup 2 --no-kernel-timing" "--steps 5 --warmup 2 --no-kernel-timing --batch 256" "--steps 5 --warmup 2 --no-kernel-timing --batch 64"; do
  python bench.py $args --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$args', '%.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], (d['roofline'] or {}).get('kernel_ms'))"
done
