#!/bin/bash
# round 4: k_mid128r with the inverse row pass of tile t and the forward row pass of tile t + 1 interleaved (-DPZ_MIDR_ILV=1) vs the product
# build, same box, alternating; parity subset under the variant first
V=${1:-ilv}
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_$V.so
echo "== parity under the variant"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "glwe or config or metric or automorphism or trace or digit or relinear or keyswitch or external" 2>&1 | tail -2
unset POULPY_HIP_LIB
for rep in 1 2; do
  bash tools/dbg/ab_libs.sh libpoulpy_hip.so variants/libpoulpy_hip_$V.so
done
echo "== key switch"
bash tools/dbg/ab_libs.sh --args "--op keyswitch" libpoulpy_hip.so variants/libpoulpy_hip_$V.so
echo "== 16 limbs key switch (32-slot tile)"
bash tools/dbg/ab_libs.sh --args "--op keyswitch --limbs 16 --batch 512" libpoulpy_hip.so variants/libpoulpy_hip_$V.so
echo "== N = 4096, 4 limbs (8-slot tile)"
bash tools/dbg/ab_libs.sh --args "--n 4096 --limbs 4 --base2k 17 --steps 100" libpoulpy_hip.so variants/libpoulpy_hip_$V.so
