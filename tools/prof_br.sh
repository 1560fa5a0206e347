#!/bin/bash
# rocprofv3 kernel statistics for the blind-rotation bench (run on the GPU box through gpurun)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_br; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in ref cbt; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$shape -- python3 $REPO/tools/bench_blind_rotation.py --shape $shape --batch 1024 --cpu-cts 0 --reps 2 > $OUT/$shape.log 2>&1
done
cd $OUT && find . -name "*kernel_trace.csv" -size +2M -delete; ls -R | head -20
