#!/bin/bash
# per-phase s_memtime totals of k_small_one (one workgroup = one ciphertext), N = 1024 and 2048, 4 limbs, 4096 per call; stamp build (-DPZ_SMALL_ONE_STAMP=1)
OUT=gpurun_out/small_one_stamps; mkdir -p $OUT
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_sostamp.so
for n in 1024 2048; do
  python bench.py --n $n --limbs 4 --base2k 17 --batch 4096 --steps 2 --warmup 1 --no-cpu-baseline --parity-samples 0 --sustained-seconds 0 --no-kernel-timing --no-margin 2>/dev/null | grep OSTAMP | sort | uniq | head -24 > $OUT/n$n.txt
  echo "== N = $n"; cat $OUT/n$n.txt | cut -c1-400 | head -10
done
unset POULPY_HIP_LIB
for n in 1024 2048; do python bench.py --n $n --limbs 4 --base2k 17 --batch 4096 --steps 50 --no-cpu-baseline --parity-samples 2 --sustained-seconds 0 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('N=$n %9.0f %s frac=%.3f %s' % (d['value'], d['unit'], d['roofline']['frac'] if d.get('roofline') else 0, (d.get('roofline') or {}).get('kernel_ms')))"; done
