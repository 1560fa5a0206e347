#!/bin/bash
# timing ablation of k_br_block_lds (results invalid): POULPY_DBG_BRL bit 0 no key loads, 1 no products, 2 no accumulator loads, 3 no LDS staging
for m in 0 1 2 4 8 9 3 15; do echo -n "brl=$m: "; POULPY_DBG_BRL=$m python tools/bench_blind_rotation.py --shape ${1:-n2048} --cpu-cts 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value']), d['kernel_classes_launches_ms'])"; done
