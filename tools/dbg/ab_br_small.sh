#!/bin/bash
# A/B library builds on the one-kernel blind rotation (N <= 1024): tools/dbg/ab_br_small.sh lib1.so lib2.so ...
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for sh in ref cbt; do
    python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-34s %-6s %8.0f /s  %s' % (sys.argv[1], sys.argv[2], d['value'], d['kernel_classes_launches_ms']))" $lib $sh
  done
done
