#!/bin/bash
# rocprofv3 kernel statistics for the circuit-bootstrapping bench and the ggsw_expand_row bench op (run on the GPU box through gpurun)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_cbt; mkdir -p $OUT
python3 $REPO/tools/bench_circuit_bootstrapping.py --batch 512 > $OUT/cbt_b512.json 2> $OUT/cbt_b512.err
python3 $REPO/tools/bench_circuit_bootstrapping.py --batch 2048 --cpu-cts 0 > $OUT/cbt_b2048.json 2> $OUT/cbt_b2048.err
python3 $REPO/bench.py --op ggsw_expand_row --no-cpu-baseline > $OUT/expand_row.json 2> $OUT/expand_row.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cbt -- python3 $REPO/tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 0 --reps 2 > $OUT/cbt.log 2>&1
cd $OUT && find . -name "*kernel_trace.csv" -size +2M -delete; ls -R | head -20
