#!/bin/bash
# round 6, GPU call 25: the 16-bit body operand for in-place calls too (flag-up fallback = a conditional i64 pre-pass + the operand variant): tests, then rates
OUT=gpurun_out/r6_run25; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "automorphism or trace or circuit or pack" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
{
for rep in 1 2; do
  python tools/dbg/auto_inplace.py 2>/dev/null | tail -1
  python tools/dbg/auto_inplace.py --out-of-place 2>/dev/null | tail -1
  python tools/dbg/auto_inplace.py --mode add 2>/dev/null | tail -1
  python tools/dbg/auto_inplace.py --mode add --out-of-place 2>/dev/null | tail -1
  python tools/dbg/auto_inplace.py --limbs 8 --batch 1024 2>/dev/null | tail -1
  python tools/dbg/auto_inplace.py --limbs 8 --batch 1024 --out-of-place 2>/dev/null | tail -1
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt | cut -c1-230
