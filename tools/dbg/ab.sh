#!/bin/bash
# A/B two library builds on the same box: parity subset + bench kernel breakdown
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  echo "== $lib"
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "glwe or config or metric or vmp" 2>&1 | tail -1
  for i in 1 2; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'], d['roofline']['kernel_ms'])"
  done
done
