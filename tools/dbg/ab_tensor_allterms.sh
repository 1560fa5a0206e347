#!/bin/bash
# round 4, item 3: k_mid_cnv3 (all three terms per tile, register-resident convolution) vs k_mid_cnv per term, same box, alternating
for rep in 1 2; do
  for v in 1 0; do
    for extra in "" "--mode square" "--relin" "--limbs 8"; do
      POULPY_DBG_TENSOR_ALLTERMS=$v python tools/bench_tensor.py --steps 10 --parity-samples 2 $extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('ALLTERMS=$v %-16s %9.0f %s  %.3f ms/step  parity=%s  %s' % ('$extra', d['value'], d['unit'], d['ms_per_step'], d.get('parity_sample',{}).get('ok'), d.get('kernel_classes_launches_ms')))"
    done
  done
done
