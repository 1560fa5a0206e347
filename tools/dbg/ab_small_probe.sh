#!/bin/bash
# same-box A/B of the small-ring kernels over library builds: tools/dbg/ab_small_probe.sh lib1.so lib2.so (relative to poulpy_amd/)
B="python bench.py --no-cpu-baseline --no-margin --steps 200"
line() { tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-42s %12.0f %-22s parity=%s | %s' % ('$1', d['value'], d['unit'], (d.get('parity_sample') or {}).get('ok'), d['config']['workload'][:70]))"; }
for rep in 1 2 3; do
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  $B --n 4096 --limbs 3 --base2k 18 2>/dev/null | line $lib
  $B --n 4096 --limbs 3 --base2k 18 --op keyswitch 2>/dev/null | line $lib
  $B --n 2048 --limbs 4 --base2k 17 2>/dev/null | line $lib
  $B --n 1024 --limbs 4 --base2k 17 --op automorphism_add 2>/dev/null | line $lib
  python tools/bench_blind_rotation.py --shape n2048 --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-42s %12.0f rotations/s n2048' % ('$lib', d['value']))"
done
done
