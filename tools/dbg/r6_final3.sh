#!/bin/bash
# round 6, final measurement, part 3 (after the add / sub and in-place 16-bit operand commits): whole GPU suite, default bench line, secondary lines
OUT=gpurun_out/r6_final3; mkdir -p $OUT; O=gpurun_out; R=06
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python bench.py > $O/r${R}_bench_default.json 2> $O/r${R}_bench_default.err
bash tools/bench_lines.sh > $O/r${R}_bench_lines.txt 2>&1; cp $O/bench_lines.jsonl $O/r${R}_bench_lines.jsonl; cp $O/tensor_lines.jsonl $O/r${R}_tensor_lines.jsonl
grep -E "automorphism|multiplications|tensorings" $O/r${R}_bench_lines.txt | cut -c1-150
grep "^{" $O/r${R}_bench_default.json | tail -1 | cut -c1-200
