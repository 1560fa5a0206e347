#!/bin/bash
# round 4, item 6: block step on the pipeline with 3 outputs per thread for 6-column shapes (k_mid128<..,BR,NCO=3>) vs the default 4
for rep in 1 2; do
for v in 1 0; do
  for args in "--shape big --batch 256" "--shape big --batch 1024" "--shape n4096"; do
    POULPY_DBG_BR_NC3=$v python tools/bench_blind_rotation.py $args --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('NC3=$v %-28s %9.0f rotations/s  %s  %s' % ('$args', d['value'], d.get('kernel_classes_launches_ms'), d.get('dispatch','')[:80]))"
  done
done
done
