#!/bin/bash
# round 6, GPU call 4: the 16-limb key-switch family on k_mid128r<4,16,..,C2> (4 ciphertexts per tile, two output-column passes) against the
# round-5 library (32-slot tile, 2 ciphertexts per key value); parity of every shape that dispatches the new form.
OUT=gpurun_out/r6_run4; mkdir -p $OUT
{
echo "== parity (configs[4] pool tests + the headline pool test: the same source serves both forms)"
timeout 1500 python -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "config5 or metric or config3 or relinear" 2>&1 | tail -3
echo "== A/B"
B="python bench.py --no-cpu-baseline --parity-samples 4 --sustained-seconds 0 --steps 20"
for rep in 1 2; do
for lib in variants/libpoulpy_hip_r5.so libpoulpy_hip.so; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for a in "--op automorphism --limbs 16 --batch 512" "--op keyswitch --limbs 16 --batch 512" "--op relinearize --limbs 16 --batch 512" "--op automorphism_add --limbs 16 --batch 512" ""; do
    $B $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('%-34s %12.0f %-22s %8.3f ms parity=%s margin=%.2g  %-48s %s' % ('$lib', d['value'], d['unit'], d['ms_per_step'], (d.get('parity_sample') or {}).get('ok'), d.get('rounding_margin') or 0, '$a', r.get('kernel_ms')))"
  done
  python tools/bench_tensor.py --parity-samples 1 --relin 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-34s %12.0f %s parity=%s %s' % ('$lib', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"
done
done
unset POULPY_HIP_LIB
python bench.py --no-cpu-baseline --parity-samples 2 --sustained-seconds 0 --steps 5 --op automorphism --limbs 16 --batch 512 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print((d.get('ceilings') or {}).get('dispatch'))"
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-260
