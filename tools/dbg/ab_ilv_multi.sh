#!/bin/bash
# several k_mid128r row-pass variants against the product build on one box: tools/dbg/ab_ilv_multi.sh ilv ilv3 ilv5 ...
LIBS="libpoulpy_hip.so"
for v in "$@"; do LIBS="$LIBS variants/libpoulpy_hip_$v.so"; done
for v in "$@"; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_$v.so
  echo "== parity under $v"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "metric or config2 or config5 or digit or relinear" 2>&1 | tail -1
done
unset POULPY_HIP_LIB
for rep in 1 2; do bash tools/dbg/ab_libs.sh $LIBS; done
echo "== key switch"; bash tools/dbg/ab_libs.sh --args "--op keyswitch" $LIBS
echo "== 16 limbs key switch (32-slot tile)"; bash tools/dbg/ab_libs.sh --args "--op keyswitch --limbs 16 --batch 512" $LIBS
