#!/bin/bash
echo "cold ubench"; timeout 120 ./tools/ubench/hbm_pass_pattern 2>&1 | grep -E "^CB 16, non|^tail, CB 16, row|one XCD per polynomial, non|mid shape \(persistent"
for i in $(seq 1 14); do python bench.py --steps 30 --no-cpu-baseline --parity-samples 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), d['roofline']['kernel_ms'])"; done
echo "warm ubench"; timeout 120 ./tools/ubench/hbm_pass_pattern 2>&1 | grep -E "^CB 16, non|^tail, CB 16, row|one XCD per polynomial, non|mid shape \(persistent"
