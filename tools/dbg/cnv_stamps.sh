#!/bin/bash
# per-phase cycle totals of k_mid_cnv3 (diagnostic build -DPZ_CNV_STAMP=1, see device_cnv.hpp): tools/dbg/cnv_stamps.sh <lib relative to poulpy_amd/> [bench_tensor args]
export POULPY_HIP_LIB=$PWD/poulpy_amd/$1; shift
python tools/bench_tensor.py --steps 1 --warmup 0 --parity-samples 0 "$@" > /tmp/cnv_stamps.log 2>&1
grep -o '"kernel_classes_launches_ms": {[^}]*}' /tmp/cnv_stamps.log | tail -1
grep CSTAMP /tmp/cnv_stamps.log | tail -32 | python -c "
import sys,re
rows=[]
for l in sys.stdin:
    m=re.match(r'CSTAMP wg (\d+) wave (\d+) tiles (\d+) total (\d+) \| (.*)', l)
    if not m: continue
    wg,wave,tiles,total=map(int,m.groups()[:4])
    d={k:int(v) for k,v in re.findall(r'([a-z0-9]+) (\d+)', m.group(5))}
    rows.append((wg,wave,tiles,total,d))
for wg,wave,tiles,total,d in sorted(rows):
    t=max(tiles,1)
    print('wg %3d wave %d tiles %3d cyc/tile %6d | ' % (wg,wave,tiles,total//t) + ' '.join('%s %5d' % (k,v//t) for k,v in d.items()))
"
