#!/bin/bash
# round 6, final measurement at HEAD: the whole GPU suite, smoke, then everything profiles/r06_* is made from (tools/measure_all.sh) + kernel resources
OUT=gpurun_out/r6_final; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
bash tools/measure_all.sh 06 > $OUT/measure_all.log 2>&1
tail -5 $OUT/measure_all.log
bash tools/prof_tensor.sh > $OUT/prof_tensor.log 2>&1
mkdir -p $OUT/prof_tensor; cp gpurun_out/prof_tensor/*.txt $OUT/prof_tensor/ 2>/dev/null
