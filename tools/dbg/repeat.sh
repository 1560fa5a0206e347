#!/bin/bash
for i in $(seq 1 12); do python bench.py --steps 30 --no-cpu-baseline --parity-samples 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), d['roofline']['kernel_ms'], d['parity_sample']['ok'])"; done
