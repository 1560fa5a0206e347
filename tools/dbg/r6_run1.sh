#!/bin/bash
# round 6, GPU call 1: tensoring A/B (round-5 library vs HEAD: product-limb window + square form in k_mid_cnv3), k_mid_cnv3 stamps,
# SQ / PMC passes of the tensoring, and same-box baselines for the small-ring shapes.
OUT=gpurun_out/r6_run1; mkdir -p $OUT
{
echo "== parity (tensoring / convolution tests, HEAD)"
timeout 1500 python -m pytest tests/test_gpu_cnv.py tests/test_gpu_scale.py -q -m gpu -x -k "tensor or cnv or convolution or relinear" 2>&1 | tail -3
echo "== A/B"
for rep in 1 2; do
for lib in variants/libpoulpy_hip_r5.so libpoulpy_hip.so; do
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
echo "== stamps: all product limbs (round-5 arithmetic), apply"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp0.so
echo "== stamps: window, apply"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp1.so
echo "== stamps: window, square form"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp1.so --mode square
echo "== stamps: all product limbs, square form"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp0.so --mode square
} > $OUT/ab.txt 2>&1
bash tools/prof_tensor.sh > $OUT/prof_tensor.log 2>&1
mkdir -p $OUT/prof_tensor && cp gpurun_out/prof_tensor/*.txt $OUT/prof_tensor/ 2>/dev/null
{
echo "== small-ring baselines (HEAD = round 5 for these kernels)"
B="python bench.py --no-cpu-baseline --parity-samples 4 --sustained-seconds 0"
for a in "--n 4096 --limbs 4 --base2k 17 --steps 100" "--n 2048 --limbs 4 --base2k 17 --steps 100" "--n 1024 --limbs 4 --base2k 17 --steps 100" \
         "--n 4096 --limbs 4 --base2k 17 --steps 100 --op keyswitch" "--n 4096 --limbs 4 --base2k 17 --steps 100 --op automorphism" \
         "--n 4096 --limbs 3 --base2k 18 --steps 100" "--op automorphism --limbs 16 --batch 512 --steps 20"; do
  $B $a 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('%12.0f %-22s %8.3f ms parity=%s  %s  %s' % (d['value'], d['unit'], d['ms_per_step'], (d.get('parity_sample') or {}).get('ok'), '$a', r.get('kernel_ms')))"
done
echo "== headline with the sustained leg"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('sustained'), (d.get('parity_sample') or {}).get('ok'))"
} > $OUT/lines.txt 2>&1
tail -5 $OUT/ab.txt
