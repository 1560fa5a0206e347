#!/bin/bash
# round 6, GPU call 17: the fused multiply + relinearize after the top-limb fix: the tensoring / relinearization tests, then per-kernel evidence of the one-call form
OUT=gpurun_out/r6_run17; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_cnv.py -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
bash tools/prof_tensor.sh --relin --one-call > $OUT/prof.log 2>&1
cp -r gpurun_out/prof_tensor $OUT/prof_tensor_onecall
cat $OUT/prof_tensor_onecall/kernel_stats.txt | cut -c1-200
