# one-kernel rotation: the four-waves-per-SIMD one-ciphertext form (build with -DPZ_BR_OCC4=1 patched in, see NOTEBOOK.md 14) against the two-ciphertext form, with and without key loads
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_exp.so
for b in 1024 2048; do for f in 2 1; do for mask in 0 16; do
  POULPY_DBG_BR_SKIP=$mask POULPY_DBG_BR_FORM=$f python tools/bench_blind_rotation.py --shape ref --batch $b --cpu-cts 0 --reps 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('batch %5d form $f skip %2d  %7.3f ms  %s' % ($b, $mask, d['ms_per_batch'], d.get('dispatch','')[:50]))"
done; done; done
