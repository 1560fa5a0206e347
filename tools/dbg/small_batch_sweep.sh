#!/bin/bash
# latency of the headline external product (and the key switch) against the batch size: ms per call and products/s at 1 ... 1024 per call
for op in external_product keyswitch; do for b in 1 2 4 8 16 32 64 128 256 1024; do
  python bench.py --op $op --batch $b --no-cpu-baseline --no-margin --parity-samples 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-18s batch %5d  %9.0f /s  %8.3f ms per call' % ('$op', $b, d['value'], d['ms_per_step']))"
done; done
