#!/bin/bash
OUT=gpurun_out/r6_run31; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -k "trace or automorphism or circuit or pack" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -2
python bench.py --no-cpu-baseline --parity-samples 2 --sustained-seconds 0 --op trace --steps 5 2>/dev/null | grep "^{" | tail -1 | cut -c1-200
