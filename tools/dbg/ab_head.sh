#!/bin/bash
# product build vs the previous commit's build (copied to poulpy_amd/variants/libpoulpy_hip_head.so before rebuilding), same box, alternating
echo "== parity (product build)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "${PARITY_K:-glwe or config or metric or automorphism or trace or digit or relinear or keyswitch or external}" 2>&1 | tail -1
LIBS="variants/libpoulpy_hip_head.so libpoulpy_hip.so"
for rep in 1 2; do bash tools/dbg/ab_libs.sh $LIBS; done
echo "== key switch"; bash tools/dbg/ab_libs.sh --args "--op keyswitch" $LIBS
echo "== 16 limbs key switch (32-slot tile)"; bash tools/dbg/ab_libs.sh --args "--op keyswitch --limbs 16 --batch 512" $LIBS
echo "== automorphism"; bash tools/dbg/ab_libs.sh --args "--op automorphism" $LIBS
echo "== N = 4096, 4 limbs (8-slot tile)"; bash tools/dbg/ab_libs.sh --args "--n 4096 --limbs 4 --base2k 17 --steps 100" $LIBS
