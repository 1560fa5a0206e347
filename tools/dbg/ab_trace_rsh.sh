#!/bin/bash
# A/B of the shifted-store tail in glwe_trace (POULPY_DBG_TRACE_RSH=0: a separate vec_znx_rsh pass before every step)
for v in 1 0 1 0; do echo -n "trace_rsh=$v: "; POULPY_DBG_TRACE_RSH=$v python bench.py --op trace --steps 5 --no-cpu-baseline --parity-samples 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'])"; done
