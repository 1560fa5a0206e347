#!/bin/bash
# round 4, item 2b: inverse inter-pass twiddle applied by the tail (-DPZ_TW_IN_TAIL=1 -DPZ_TW_TAIL_REGS=1) vs the product build, same box:
# parity subset under the variant, then bench lines alternating the two libraries
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_twtail.so
echo "== parity under the variant (glwe / metric / config / automorphism / trace tests)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "glwe or config or metric or automorphism or trace or digit" 2>&1 | tail -2
unset POULPY_HIP_LIB
for rep in 1 2; do
  bash tools/dbg/ab_libs.sh libpoulpy_hip.so variants/libpoulpy_hip_twtail.so
done
echo "== key switch"
bash tools/dbg/ab_libs.sh --args "--op keyswitch" libpoulpy_hip.so variants/libpoulpy_hip_twtail.so
