for rep in 1 2; do for lib in variants/libpoulpy_hip_head.so libpoulpy_hip.so; do for sh in ref cbt; do for b in 256 512 1024; do
POULPY_HIP_LIB=$PWD/poulpy_amd/$lib python tools/bench_blind_rotation.py --shape $sh --batch $b --cpu-cts 0 --reps 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-34s %-6s batch %5d %9.0f rotations/s  %7.3f ms  %s' % ('$lib', d['shape'], d['batch'], d['value'], d['ms_per_batch'], d.get('dispatch','')[:58]))"
done; done; done; done
