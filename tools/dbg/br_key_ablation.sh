export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
for sh in ref cbt; do for mask in 0 16 2 18; do
  POULPY_DBG_BR_SKIP=$mask POULPY_DBG_BR_FORM=2 python tools/bench_blind_rotation.py --shape $sh --batch 512 --cpu-cts 0 --reps 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-6s skip %2d  %7.3f ms  %s' % ('$sh', $mask, d['ms_per_batch'], d.get('dispatch','')[:60]))"
done; done
