#!/bin/bash
# round 6, GPU call 11: k_small_one with the loads of both column-pass sweeps and the first key row requested ahead, against the first version (variants/libpoulpy_hip_one1.so)
OUT=gpurun_out/r6_run11; mkdir -p $OUT
{
echo "== parity"
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "small or n1024 or n2048 or config1 or pool or sweep or grid or external or keyswitch" 2>&1 | grep -E "passed|failed|rror" | tail -3
echo "== A/B"
B="python bench.py --no-cpu-baseline --parity-samples 4 --sustained-seconds 0 --steps 200"
for rep in 1 2 3; do
for lib in variants/libpoulpy_hip_one1.so libpoulpy_hip.so; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for a in "--n 1024 --limbs 4 --base2k 17" "--n 2048 --limbs 4 --base2k 17" "--n 1024 --limbs 4 --base2k 17 --op keyswitch" "--n 2048 --limbs 3 --base2k 18" "--n 1024 --limbs 2 --base2k 20"; do
    $B $a 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('%-34s %12.0f %-22s %8.4f ms parity=%s  %-48s %s' % ('$lib', d['value'], d['unit'], d['ms_per_step'], (d.get('parity_sample') or {}).get('ok'), '$a', r.get('kernel_ms')))"
  done
done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
