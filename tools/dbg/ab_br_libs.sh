#!/bin/bash
# blind rotation / circuit bootstrapping with several library builds on one box: tools/dbg/ab_br_libs.sh lib1.so lib2.so ... (relative to poulpy_amd/)
echo "== parity (blind rotation / LWE / circuit bootstrapping tests, product build)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lwe.py tests/test_gpu_scale.py -q -m gpu -x -k "blind or circuit or lwe or pack or bootstrap" 2>&1 | tail -1
for rep in 1 2; do
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  IFS=';' read -ra SHAPES <<< "${BR_SHAPES:---shape n2048;--shape cbt;--shape ref;--shape n4096}"
  for args in "${SHAPES[@]}"; do
    python tools/bench_blind_rotation.py $args --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-34s %-16s %9.0f rotations/s  %s' % ('$lib', '$args', d['value'], d.get('kernel_classes_launches_ms')))"
  done
  [ -n "$BR_NO_CBT" ] || python tools/bench_circuit_bootstrapping.py --batch 512 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-34s %-16s %9.0f bootstrappings/s  %s' % ('$lib', 'circuit', d['value'], d.get('kernel_classes_launches_ms')))"
done
done
