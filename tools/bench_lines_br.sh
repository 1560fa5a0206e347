#!/bin/bash
# blind-rotation / circuit-bootstrapping lines at HEAD (GPU box): one JSON line each into gpurun_out/bench_lines_br.jsonl
OUT=gpurun_out/bench_lines_br.jsonl; : > $OUT
for sh in ref cbt n2048 n4096; do python tools/bench_blind_rotation.py --shape $sh --cpu-cts 0 2>/dev/null | tail -1 >> $OUT; done
python tools/bench_blind_rotation.py --shape big --batch 1024 --cpu-cts 0 2>/dev/null | tail -1 >> $OUT
python tools/bench_circuit_bootstrapping.py --batch 512 2>/dev/null | tail -1 >> $OUT
python tools/bench_circuit_bootstrapping.py --batch 1024 2>/dev/null | tail -1 >> $OUT
cut -c1-400 $OUT
