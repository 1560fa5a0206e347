for m in 0 1 2 3 4 7; do echo -n "skip=$m: "; POULPY_DBG_SMALL_SKIP=$m python bench.py --n 4096 --limbs 4 --base2k 17 --steps 30 --no-cpu-baseline --parity-samples 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), d['roofline']['kernel_ms'])"; done
