#!/bin/bash
for d in 0 1 2 3 4 8 12 15; do
  POULPY_MID_DBG=$d python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg', $d, 'fused_mid ms', d['roofline']['kernel_ms'].get('fused_mid'))"
done
