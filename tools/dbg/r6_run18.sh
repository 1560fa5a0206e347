#!/bin/bash
# round 6, GPU call 18: one-call multiply + relinearize: phase between the 16-bit tensor columns, f64 steps in the pairwise tail (variant), vs two calls
OUT=gpurun_out/r6_run18; mkdir -p $OUT
line() { python tools/bench_tensor.py --parity-samples 1 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-34s %-40s %8.0f %s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"; }
{
for rep in 1 2; do
  line "two calls" "--relin"
  for ph in 0 256 768 1536 2560; do
    export POULPY_DBG_T16_PHASE_KIB=$ph
    line "phase=$ph" "--relin --one-call"
  done
  unset POULPY_DBG_T16_PHASE_KIB
  export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_d16rf64.so
  line "f64 pairwise (variant)" "--relin --one-call"
  line "f64 pairwise (variant)" ""
  line "f64 pairwise (variant)" "--mode square"
  unset POULPY_HIP_LIB
  line "HEAD" ""
  line "HEAD" "--mode square"
  line "HEAD" "--relin --one-call --mode square"
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-230
