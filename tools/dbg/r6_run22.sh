#!/bin/bash
# round 6, GPU call 22: plain spectral automorphism with the body as 16-bit copies on the f64 chain (experiment build: POULPY_DBG_AUTO_BODY16=0/1),
# and the phase between the 16-bit tensor columns of the one-call multiplication (POULPY_DBG_T16_PHASE_KIB) - both knobs need -DPZ_EXPERIMENT
OUT=gpurun_out/r6_run22; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "automorphism or trace or circuit" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
line() { python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 2 --timing-steps 10 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d.get('roofline') or {}
print('%-9s %-58s %9.0f %-18s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], (d.get('parity_sample') or {}).get('ok'), r.get('kernel_ms')))"; }
tline() { python tools/bench_tensor.py --parity-samples 1 $2 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-12s %-40s %8.0f %s parity=%s %s' % ('$1', '$2', d['value'], d['unit'], d['parity_ok'], d['kernel_classes_launches_ms']))"; }
{
for rep in 1 2 3; do
  for v in 0 1; do
    export POULPY_DBG_AUTO_BODY16=$v
    line body16=$v "--op automorphism --limbs 16 --batch 512 --steps 20"
    line body16=$v "--op automorphism"
    line body16=$v "--op automorphism --galois 1979 --limbs 16 --batch 512 --steps 20"
  done
  unset POULPY_DBG_AUTO_BODY16
done
for rep in 1 2; do
  for ph in 0 768 2560; do
    export POULPY_DBG_T16_PHASE_KIB=$ph
    tline "phase=$ph" "--relin --one-call"
  done
  unset POULPY_DBG_T16_PHASE_KIB
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-250
