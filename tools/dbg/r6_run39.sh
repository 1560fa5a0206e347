#!/bin/bash
OUT=gpurun_out/r6_run39; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "body_as_16_bit" > $OUT/pytest.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest.txt | tail -3
