#!/bin/bash
# the GPU parity suites once more with HIP graphs off (POULPY_DBG_GRAPHS=0): the round's new paths must not depend on replay
OUT=gpurun_out/r6_run41; mkdir -p $OUT
POULPY_DBG_GRAPHS=0 timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cnv.py tests/test_gpu_lwe.py tests/test_gpu_structured.py -x -q -m gpu > $OUT/pytest_nographs.txt 2>&1
grep -E "passed|failed" $OUT/pytest_nographs.txt | tail -2
