#!/bin/bash
# round 4, blind rotation (BASELINE configs[3]): lines at HEAD incl. the large ring at batch 1024 (configs[3] shards 8192 over 8 GPUs: 1024 each),
# and s_memtime stamps of the block step on the pipeline (k_mid128<..,BR>, diagnostic build) at N = 2^14
for sh in ref cbt n2048 n4096; do python tools/bench_blind_rotation.py --shape $sh --cpu-cts 0 2>/dev/null | tail -1 | cut -c1-330; done
for b in 256 1024; do python tools/bench_blind_rotation.py --shape big --batch $b --cpu-cts 0 2>/dev/null | tail -1 | cut -c1-330; done
python tools/bench_circuit_bootstrapping.py --batch 512 2>/dev/null | tail -1 | cut -c1-330
echo "== stamps, N = 2^14, 2 blocks, batch 256 / 1024"
for b in 256 1024; do
POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_stamp.so python tools/bench_blind_rotation.py --shape big --batch $b --reps 1 --n-lwe 14 --cpu-cts 0 2>&1 | grep STAMP | tail -16 | python -c "
import sys,re
for l in sys.stdin:
    m=re.match(r'STAMP wg (\d+) wave (\d+) tiles (\d+) total (\d+) \| (.*)', l)
    if not m: continue
    wg,wave,tiles,total=map(int,m.groups()[:4])
    d={k:int(v) for k,v in re.findall(r'([a-z0-9]+) (\d+)', m.group(5))}
    t=max(tiles,1)
    print('wg %3d wave %d tiles %3d cyc/tile %6d | ' % (wg,wave,tiles,total//t) + ' '.join('%s %5d' % (k,v//t) for k,v in d.items()))
"
done
