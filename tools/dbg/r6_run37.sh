#!/bin/bash
# round 6, GPU call 37: small-ring blind rotation (N = 2048; circuit bootstrapping's rotation) with 16-bit accumulator digits between blocks (HEAD) vs 32-bit (POULPY_DBG_BR_ACC16=0, experiment build)
OUT=gpurun_out/r6_run37; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -k "blind or rotation or bootstrap or lwe" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -2
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
one() { python tools/bench_blind_rotation.py --shape $2 --batch 1024 --cpu-cts 2 --reps 3 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-6s %-6s %9.0f %s parity=%s %s' % ('$1', '$2', d['value'], d.get('unit','')[:12], d.get('parity_on_cpu_sample'), d.get('kernel_classes_launches_ms')))"; }
cbt() { python tools/bench_circuit_bootstrapping.py --batch 1024 --cpu-cts 1 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-6s %-6s %9.0f %s parity=%s %s' % ('$1', 'cbt', d['value'], d.get('unit','')[:14], d.get('parity_on_cpu_sample'), d.get('kernel_classes_launches_ms')))"; }
{
for rep in 1 2; do
for v in acc32 acc16; do
  unset POULPY_DBG_BR_ACC16; [ $v = acc32 ] && export POULPY_DBG_BR_ACC16=0
  one $v n2048
  cbt $v
done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-260
