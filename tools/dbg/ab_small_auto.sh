for op in automorphism automorphism_add; do for v in 1 0; do echo -n "$op small_auto=$v: "; POULPY_DBG_SMALL_AUTO=$v python bench.py --n 4096 --limbs 3 --base2k 18 --op $op --steps 50 --no-cpu-baseline --parity-samples 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), d['roofline']['kernel_ms'], d['parity_sample']['ok'])"; done; done
for n in 2048 1024; do for v in 1 0; do echo -n "n=$n automorphism_add small_auto=$v: "; POULPY_DBG_SMALL_AUTO=$v python bench.py --n $n --limbs 4 --base2k 17 --op automorphism_add --steps 50 --no-cpu-baseline --parity-samples 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), d['roofline']['kernel_ms'], d['parity_sample']['ok'])"; done; done
