#!/bin/bash
# per-phase cycle totals of k_small_inv (diagnostic build -DPZ_SMALL_STAMP=1): tools/dbg/small_stamps.sh <lib relative to poulpy_amd/> [bench args]
export POULPY_HIP_LIB=$PWD/poulpy_amd/$1; shift
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --parity-samples 0 --timing-steps 1 "$@" > /tmp/small_stamps.log 2>&1
grep -o '"kernel_ms": {[^}]*}' /tmp/small_stamps.log | tail -1
grep SSTAMP /tmp/small_stamps.log | tail -16 | sort -k5n
