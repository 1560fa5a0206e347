#!/bin/bash
# small-ring shapes (N <= 4096): GLWE ops through bench.py, blind rotation / circuit bootstrapping, the per-op transforms
B="python bench.py --no-cpu-baseline"
line() { tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%12.0f %-22s parity=%s %s | %s' % (d['value'], d['unit'], d.get('parity_ok', d.get('parity')), d['roofline'].get('kernel_ms', ''), d['config']['workload'][:90]))"; }
$B --n 4096 --limbs 4 --base2k 17 2>/dev/null | line
POULPY_DBG_SMALL=2 $B --n 4096 --limbs 4 --base2k 17 2>/dev/null | line
$B --n 4096 --limbs 3 --base2k 18 2>/dev/null | line
$B --n 4096 --limbs 3 --base2k 18 --op keyswitch 2>/dev/null | line
$B --n 4096 --limbs 3 --base2k 18 --op automorphism 2>/dev/null | line
$B --n 2048 --limbs 4 --base2k 17 2>/dev/null | line
$B --n 1024 --limbs 4 --base2k 17 2>/dev/null | line
$B --n 4096 --limbs 4 --base2k 17 --op trace 2>/dev/null | line
for ln in 10 11 12; do
  for op in dft idft; do
    python tools/bench_hal_ops.py --op $op --n $((1 << ln)) --cols 1 --limbs 1 --batch $(( (1 << 27) >> ln )) 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%12.0f %-22s frac %.3f %s' % (d['value'], d['unit'], d['roofline']['frac'], d['config']['workload'][:80]))"
  done
done
for sh in ref cbt n2048; do python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%12.0f %-22s parity=%s %s' % (d['value'], d.get('unit', d.get('metric')), d.get('parity_ok', d.get('parity')), str(d.get('config', {}).get('workload', d.get('config')))[:80]))"; done
python tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%12.0f %-22s parity=%s %s' % (d['value'], d.get('unit', d.get('metric')), d.get('parity_ok', d.get('parity')), str(d.get('config', {}).get('workload', d.get('config')))[:80]))"
