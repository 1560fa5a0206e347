#!/bin/bash
# plain glwe_automorphism: spectral form (default) vs key switch + signed permutation pass (POULPY_DBG_AUTO_SPECTRAL=2)
for v in 1 2 1 2; do for l in 8 16; do echo -n "auto_spectral=$v limbs=$l: "; b=$((8192/l)); POULPY_DBG_AUTO_SPECTRAL=$v python bench.py --op automorphism --limbs $l --batch $b --steps 30 --no-cpu-baseline --parity-samples 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), d['roofline']['kernel_ms'], d['parity_sample']['ok'])"; done; done
