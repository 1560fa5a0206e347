#!/bin/bash
# everything profiles/rNN_* is made from, in one call on the GPU box (about 10 minutes): tools/measure_all.sh NN
R=${1:-05}; O=gpurun_out
python bench.py > $O/r${R}_bench_default.json 2> $O/r${R}_bench_default.err
bash tools/bench_lines.sh > $O/r${R}_bench_lines.txt 2>&1; cp $O/bench_lines.jsonl $O/r${R}_bench_lines.jsonl; cp $O/tensor_lines.jsonl $O/r${R}_tensor_lines.jsonl
bash tools/bench_lines_br.sh > $O/r${R}_bench_lines_br.txt 2>&1; cp $O/bench_lines_br.jsonl $O/r${R}_bench_lines_br.jsonl
bash tools/bench_lines_hal.sh > $O/r${R}_bench_lines_hal.txt 2>&1; cp $O/bench_lines_hal.jsonl $O/r${R}_bench_lines_hal.jsonl
bash tools/prof.sh > $O/r${R}_prof.log 2>&1
cp $O/prof/trace_steady.txt $O/r${R}_kernel_steady.txt 2>/dev/null
find $O/prof/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r${R}_kernel_stats.csv
cp $O/prof/pmc_fetch_summary.txt $O/r${R}_pmc_fetch_summary.txt; cp $O/prof/pmc_write_summary.txt $O/r${R}_pmc_write_summary.txt
python tools/traffic_json.py $O/r${R}_pmc_fetch_summary.txt $O/r${R}_pmc_write_summary.txt $O/r${R}_traffic.json 1024 > $O/r${R}_traffic.txt 2>&1   # (with the kernels' code-object signatures)
ls -la $O | grep r${R}_ | head -30
