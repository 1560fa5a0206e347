#!/bin/bash
# round 6, GPU call 14: static priority for the half-pairwise waves during k_mid_cnv3's convolution (-DPZ_CNV_PRIO=2) vs HEAD
OUT=gpurun_out/r6_run14; mkdir -p $OUT
{
for rep in 1 2 3; do
for lib in libpoulpy_hip.so variants/libpoulpy_hip_cnvprio.so; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for args in "" "--relin" "--mode square" "--limbs 8 --batch 512"; do
    python tools/bench_tensor.py --parity-samples 1 $args 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-36s %-22s %8.0f %s parity=%s %s' % ('$lib', '$args', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"
  done
done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-220
