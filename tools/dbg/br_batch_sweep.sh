#!/bin/bash
# blind rotation / circuit bootstrapping against the batch size (product library): ms per call and rate; looks for steps in the curve
for sh in ${1:-n2048 n4096}; do for b in ${2:-16 64 128 256 384 512 768 1024}; do
  python tools/bench_blind_rotation.py --shape $sh --batch $b --cpu-cts 0 --reps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-8s batch %5d %9.0f rotations/s  %8.3f ms' % (d['shape'], d['batch'], d['value'], d['ms_per_batch']))"
done; done
for b in ${3:-16 64 128 256 512 1024}; do
  python tools/bench_circuit_bootstrapping.py --batch $b --cpu-cts 0 --reps 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-8s batch %5d %9.0f bootstrappings/s  %8.3f ms' % ('circuit', d['batch'], d['value'], d.get('ms_per_batch', 0)))"
done
