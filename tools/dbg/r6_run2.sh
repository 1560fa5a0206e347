#!/bin/bash
# round 6, GPU call 2: tensoring with the static product-limb window + early operand loads (k_mid_cnv3) and the 16-bit side copies of the diagonal
# digits (tails) against the round-5 library; stamps; structured-input margins.
OUT=gpurun_out/r6_run2; mkdir -p $OUT
{
echo "== parity (tensoring / convolution tests + the new host-path test, HEAD)"
timeout 1500 python -m pytest tests/test_gpu_cnv.py tests/test_gpu_scale.py tests/test_gpu_parity.py -q -m gpu -x -k "tensor or cnv or convolution or relinear or pinned_host" 2>&1 | tail -3
echo "== A/B"
for rep in 1 2; do
for lib in variants/libpoulpy_hip_r5.so variants/libpoulpy_hip_noearly.so libpoulpy_hip.so; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for args in "" "--relin" "--mode square" "--limbs 8 --batch 512"; do
    python tools/bench_tensor.py --parity-samples 1 $args 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-36s %-22s %8.0f %s parity=%s margin=%.2g %s' % ('$lib', '$args', d['value'], d['unit'], d['parity_ok'], d['rounding_margin'] or 0, d['kernel_classes_launches_ms']))"
  done
done
done
unset POULPY_HIP_LIB
echo "== stamps: HEAD, apply"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp1.so | head -12
echo "== stamps: HEAD, square"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp1.so --mode square | head -12
} > $OUT/ab.txt 2>&1
python tools/margin_structured.py --out $OUT/margin_structured.md > $OUT/margin_structured.log 2>&1
tail -5 $OUT/ab.txt; tail -8 $OUT/margin_structured.log
