#!/bin/bash
# round 6, GPU call 7: is the structured blind-rotation failure at N = 2^14 (batch 2, two blocks) older than round 6?  The same tests on the round-5 library,
# on HEAD, on HEAD without HIP graphs, on HEAD with workspace canaries - fresh processes, twice.
OUT=gpurun_out/r6_run7; mkdir -p $OUT
{
for rep in 1 2; do
for cfg in "POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_r5.so" "X=1" "POULPY_DBG_GRAPHS=0" "POULPY_DBG_CANARY=1"; do
  echo "== $cfg"
  env $cfg timeout 600 python -m pytest tests/test_gpu_structured.py -q -m gpu -k "br_big or external" 2>&1 | grep -E "passed|failed|^FAILED|gpu_margin" | tail -8 | cut -c1-260
done
done
echo "== br_big only, HEAD"
timeout 600 python -m pytest tests/test_gpu_structured.py -q -m gpu -k "br_big" 2>&1 | grep -E "passed|failed|^FAILED|gpu_margin" | tail -8 | cut -c1-260
} > $OUT/structured.txt 2>&1
cat $OUT/structured.txt
