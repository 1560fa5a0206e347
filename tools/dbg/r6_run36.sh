#!/bin/bash
OUT=gpurun_out/r6_run36; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -k "small or one_kernel or ring or n2048 or 2048 or fused_tail_shapes or external_product" > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -2
python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 8 --steps 100 --n 2048 --limbs 2 --base2k 17 2>/dev/null | grep "^{" | tail -1 | cut -c1-160
python bench.py --no-cpu-baseline --sustained-seconds 0 --parity-samples 8 --steps 100 --n 2048 --limbs 2 --base2k 17 --op keyswitch 2>/dev/null | grep "^{" | tail -1 | cut -c1-160
