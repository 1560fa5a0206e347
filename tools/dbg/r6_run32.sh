#!/bin/bash
# round 6, GPU call 32: blind rotation on the pipeline path (N = 4096, 2^14) with the accumulator between two blocks as 16-bit tile-order digits (HEAD) vs
# 32-bit natural order (POULPY_DBG_BR_ACC16=0, experiment build); BR / LWE / circuit tests first
OUT=gpurun_out/r6_run32; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -k "blind or rotation or bootstrap or lwe or structured" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -2
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
one() { python tools/bench_blind_rotation.py --shape $2 --batch 1024 --cpu-cts 2 --reps 2 $3 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); gb=d.get('gate_bootstrap') or {}
print('%-6s %-6s %-18s %9.0f %s parity=%s margin=%.2g %s gate=%s' % ('$1', '$2', '$3', d['value'], d.get('unit','')[:12], d.get('parity_on_cpu_sample'), d.get('rounding_margin') or 0, d.get('kernel_classes_launches_ms'), gb.get('gate_bootstraps_per_s')))"; }
{
for rep in 1 2; do
for v in acc32 acc16; do
  unset POULPY_DBG_BR_ACC16; [ $v = acc32 ] && export POULPY_DBG_BR_ACC16=0
  one $v n4096 ""
  one $v big ""
done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-260
