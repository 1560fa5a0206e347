#!/bin/bash
OUT=gpurun_out/r6_run28; mkdir -p $OUT
python tools/dbg/trace_limbs.py 8192 full > $OUT/f8192.txt 2>&1; grep FULL $OUT/f8192.txt
python tools/dbg/trace_limbs.py 65536 full > $OUT/f65536.txt 2>&1; grep FULL $OUT/f65536.txt; tail -3 $OUT/f65536.txt | cut -c1-200
