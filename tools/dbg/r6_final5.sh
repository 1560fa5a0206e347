#!/bin/bash
# round 6, final measurement, part 5 (after the blind rotation's 16-bit accumulator): whole GPU suite, smoke, the blind-rotation / gate-bootstrap / circuit-bootstrapping lines
OUT=gpurun_out/r6_final5; mkdir -p $OUT; O=gpurun_out; R=06
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
bash tools/bench_lines_br.sh > $O/r${R}_bench_lines_br.txt 2>&1; cp $O/bench_lines_br.jsonl $O/r${R}_bench_lines_br.jsonl
cat $O/r${R}_bench_lines_br.txt | cut -c1-200
