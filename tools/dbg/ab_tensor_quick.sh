#!/bin/bash
# same-box A/B of the tensoring lines over library builds: tools/dbg/ab_tensor_quick.sh lib1.so lib2.so (relative to poulpy_amd/)
for rep in 1 2 3; do
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for args in "" "--relin" "--mode square" "--limbs 8 --batch 512"; do
    python tools/bench_tensor.py $args 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-36s %-22s %8.0f %s parity=%s %s' % ('$lib', '$args', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"
  done
done
done
