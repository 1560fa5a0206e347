#!/bin/bash
for rep in 1 2 3; do
for plan in t w; do
  for op in external_product keyswitch; do
    POULPY_DBG_SPLIT=$plan python bench.py --op $op --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plan', '$plan', '$op', '%.0f' % d['value'], d['roofline']['kernel_ms'])"
  done
done
done
