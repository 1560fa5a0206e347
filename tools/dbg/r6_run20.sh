#!/bin/bash
# round 6, GPU call 20: side-copy tensoring tails with "normalizing, never raw" known at compile time (HEAD) vs the run-time tests (nzd0 = previous HEAD),
# and the f64 steps in the pairwise side-copy tails on top of it (d16rf64); tensoring tests first
OUT=gpurun_out/r6_run20; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_cnv.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_d16rf64.so timeout 1500 python -m pytest tests/test_gpu_cnv.py -x -q -m gpu -k "tensor" > $OUT/pytest_f64.txt 2>&1
tail -3 $OUT/pytest_f64.txt
line() { python tools/bench_tensor.py --parity-samples 1 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-10s %-40s %8.0f %s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"; }
{
for rep in 1 2 3; do
  for v in nzd0 head d16rf64; do
    if [ $v = head ]; then unset POULPY_HIP_LIB; else export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_$v.so; fi
    line $v ""
    line $v "--mode square"
    line $v "--relin --one-call"
    line $v "--limbs 8 --batch 512"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-230
