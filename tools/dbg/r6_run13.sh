#!/bin/bash
# round 6, GPU call 13: the N = 2048 key switch (4 input polynomials) on k_small_one (variant -DPZ_SMALL_ONE_ALL=1) vs the two-kernel pipeline (HEAD); the new pool test
OUT=gpurun_out/r6_run13; mkdir -p $OUT
{
echo "== pool test of the one-kernel product + structured tests"
timeout 1700 python -m pytest tests/test_gpu_scale.py tests/test_gpu_structured.py -q -m gpu -x -k "one_kernel_product or structured" 2>&1 | grep -E "passed|failed|rror" | tail -3
echo "== A/B"
B="python bench.py --no-cpu-baseline --parity-samples 4 --sustained-seconds 0 --steps 200"
for rep in 1 2 3; do
for lib in libpoulpy_hip.so variants/libpoulpy_hip_oneall.so; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for a in "--n 2048 --limbs 4 --base2k 17 --op keyswitch" "--n 2048 --limbs 3 --base2k 18 --op keyswitch" "--n 2048 --limbs 4 --base2k 17 --op keyswitch --batch 4096"; do
    $B $a 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('%-34s %12.0f %-22s %8.4f ms parity=%s  %-58s %s' % ('$lib', d['value'], d['unit'], d['ms_per_step'], (d.get('parity_sample') or {}).get('ok'), '$a', r.get('kernel_ms')))"
  done
done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
