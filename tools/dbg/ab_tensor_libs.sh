#!/bin/bash
# GLWE tensoring lines with several library builds on one box: tools/dbg/ab_tensor_libs.sh lib1.so lib2.so ... (relative to poulpy_amd/)
echo "== parity (tensoring / convolution tests, product build)"
timeout 1500 python -m pytest tests/test_gpu_cnv.py tests/test_gpu_scale.py -q -m gpu -x -k "tensor or cnv or convolution or relinear" 2>&1 | tail -1
for rep in 1 2; do
for lib in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for args in "" "--mode square" "--relin" "--limbs 8 --batch 512"; do
    python tools/bench_tensor.py $args 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-34s %-22s %9.0f %s  %s' % ('$lib', '$args', d['value'], d['unit'], d.get('kernel_classes_launches_ms')))"
  done
done
done
