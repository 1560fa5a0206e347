#!/bin/bash
# round 6, GPU call 16: the fused multiply + relinearize (tensor as 16-bit digits in scratch): parity tests, then one call vs two calls
OUT=gpurun_out/r6_run16; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_cnv.py -x -q -m gpu -k "mul_relinearize or tensor_relinearize or tensor_apply_fused or n65536" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
{
for rep in 1 2 3; do
  for args in "--relin" "--relin --one-call" "--relin --mode square" "--relin --one-call --mode square" "--relin --limbs 8 --batch 512" "--relin --one-call --limbs 8 --batch 512"; do
    POULPY_DBG_DISPATCH=1 python tools/bench_tensor.py --parity-samples 1 $args 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-44s %8.0f %s parity=%s %s' % ('$args', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-220
