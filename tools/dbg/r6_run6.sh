#!/bin/bash
# round 6, GPU call 6: the one-kernel GLWE product at N = 1024 / 2048 (k_small_one) against the round-5 library (two-kernel pipeline).
OUT=gpurun_out/r6_run6; mkdir -p $OUT
{
echo "== parity: everything that runs at N <= 2048 plus the pool tests"
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_margin.py tests/test_gpu_lwe.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -5
echo "== A/B"
B="python bench.py --no-cpu-baseline --parity-samples 4 --sustained-seconds 0 --steps 100"
for rep in 1 2; do
for lib in variants/libpoulpy_hip_r5.so libpoulpy_hip.so; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for a in "--n 1024 --limbs 4 --base2k 17" "--n 2048 --limbs 4 --base2k 17" "--n 1024 --limbs 4 --base2k 17 --op keyswitch" "--n 2048 --limbs 4 --base2k 17 --op keyswitch" "--n 2048 --limbs 3 --base2k 18" "--n 1024 --limbs 2 --base2k 20" "--n 2048 --limbs 4 --base2k 17 --batch 4096" "--n 1024 --limbs 4 --base2k 17 --batch 4096"; do
    $B $a 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('%-34s %12.0f %-22s %8.4f ms parity=%s margin=%.2g hbm=%.3f  %-48s %s' % ('$lib', d['value'], d['unit'], d['ms_per_step'], (d.get('parity_sample') or {}).get('ok'), d.get('rounding_margin') or 0, (d.get('ceilings') or {}).get('hbm',{}).get('frac',0) if isinstance((d.get('ceilings') or {}).get('hbm'), dict) else 0, '$a', r.get('kernel_ms')))"
  done
done
done
unset POULPY_HIP_LIB
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-300
{
echo "== structured tests alone"
timeout 900 python -m pytest tests/test_gpu_structured.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|gpu_margin" | tail -12 | cut -c1-300
echo "== structured blind-rotation cases, twice, fresh process"
python - <<'PY'
import sys; sys.path.insert(0, "tools")
import margin_structured as ms
from oracle.ref import RefModule
from poulpy_amd.hal import Module
for rep in range(2):
    hip, ref = Module(16384, device=0), RefModule(16384)
    for name in ("min", "alt", "tone:1", "tone:N/4", "tone:N/2-1", "delta", "uniform"):
        r = ms.SHAPES["br_big"][1](hip, ref, 13, name)
        print(rep, name, r, flush=True)
PY
} > $OUT/structured.txt 2>&1
cat $OUT/structured.txt | cut -c1-300
