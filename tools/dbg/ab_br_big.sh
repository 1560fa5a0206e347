#!/bin/bash
# A/B library builds on the blind-rotation shapes that run the block step on the pipeline (N >= 4096): tools/dbg/ab_br_big.sh lib1.so lib2.so ...
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  python tools/bench_blind_rotation.py --shape n4096 --batch 1024 --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-34s %-6s %8.0f /s  parity=%s %s' % (sys.argv[1], 'n4096', d['value'], d.get('parity_ok', d.get('parity')), d['kernel_classes_launches_ms']))" $lib
  python tools/bench_blind_rotation.py --shape big --batch 256 --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-34s %-6s %8.0f /s  parity=%s %s' % (sys.argv[1], 'big', d['value'], d.get('parity_ok', d.get('parity')), d['kernel_classes_launches_ms']))" $lib
done
