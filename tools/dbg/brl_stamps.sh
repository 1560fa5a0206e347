#!/bin/bash
# per-phase s_memtime totals of k_br_block_lds (diagnostic build: POULPY_BUILD_DEFS=-DPZ_BRL_STAMP=1 POULPY_BUILD_TAG=brlstamp), second block of a rotation
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_brlstamp.so
for sh in ${BRL_SHAPES:-n2048}; do
  echo "== $sh"
  python tools/bench_blind_rotation.py --shape $sh --cpu-cts 0 --reps 1 --n-lwe 21 2>&1 | grep BSTAMP | sort -u | head -24
done
echo "== circuit bootstrapping (k_br_block_lds<2,9,3>)"
python tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 0 --reps 1 --n-lwe 21 2>&1 | grep BSTAMP | sort -u | head -24
