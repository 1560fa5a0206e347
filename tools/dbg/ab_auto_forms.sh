#!/bin/bash
# round 4, late: the tail's operand forms as straight-line loops (product build) vs the per-element form of the commits before (head), and on
# top the two knobs that failed in that form: POULPY_DBG_AUTO_BODYADD=1 (pre-pass only permutes, the tail adds a0), POULPY_DBG_AUTO_FOLD=1
# (the tail gathers phi(body) itself)
echo "== parity (product build; then with each knob)"
for env in "X=0" "POULPY_DBG_AUTO_BODYADD=1" "POULPY_DBG_AUTO_FOLD=1"; do
  env $env timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -m gpu -x -k "automorphism or trace or config5 or rotate or pack or circuit" 2>&1 | tail -1
done
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --parity-samples 2"
show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-52s %9.0f /s  parity=%s  %s' % ('$1', d['value'], (d.get('parity_sample') or {}).get('ok'), {k: round(v,3) for k,v in d['roofline'].get('kernel_ms',{}).items()}))"; }
for rep in 1 2; do
for op in automorphism_add automorphism; do
  for g in 5 78125; do
    POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_head.so $B --op $op --galois $g 2>/dev/null | show "$op g=$g head"
    $B --op $op --galois $g 2>/dev/null | show "$op g=$g product"
    POULPY_DBG_AUTO_BODYADD=1 $B --op $op --galois $g 2>/dev/null | show "$op g=$g product bodyadd=1"
    POULPY_DBG_AUTO_FOLD=1 $B --op $op --galois $g 2>/dev/null | show "$op g=$g product fold=1"
  done
done
done
echo "== trace, key switch, relinearize (unchanged paths)"
for lib in variants/libpoulpy_hip_head.so libpoulpy_hip.so; do
  POULPY_HIP_LIB=$PWD/poulpy_amd/$lib $B --op trace --steps 5 2>/dev/null | show "trace $lib"
  POULPY_HIP_LIB=$PWD/poulpy_amd/$lib $B --op keyswitch 2>/dev/null | show "keyswitch $lib"
  POULPY_HIP_LIB=$PWD/poulpy_amd/$lib $B --op relinearize 2>/dev/null | show "relinearize $lib"
done
