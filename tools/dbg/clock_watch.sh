#!/bin/bash
# samples GPU clock / power with rocm-smi while a bench runs: tools/dbg/clock_watch.sh <bench args>
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --json 2>/dev/null | python -c "
import sys,json
try:
    d=json.load(sys.stdin); c=d.get('card0',{})
    print({k:v for k,v in c.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'mclk' in k.lower() or 'fclk' in k.lower()})
except Exception as e: print('err',e)
"; sleep 0.25; done ) > gpurun_out/clock_watch.txt 2>&1 &
W=$!
python bench.py --steps 400 --warmup 3 --no-cpu-baseline --parity-samples 0 "$@" 2>/dev/null | tail -1 | cut -c1-200
wait $W
sort gpurun_out/clock_watch.txt | uniq -c | sort -rn | head -12
