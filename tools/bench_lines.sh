#!/bin/bash
# secondary bench.py lines at HEAD (run on the GPU box through gpurun): one JSON line each into gpurun_out/bench_lines.jsonl
OUT=gpurun_out/bench_lines.jsonl; : > $OUT
B="python bench.py --no-cpu-baseline --parity-samples 4 --steps 30 --sustained-seconds 0"   # (the sustained leg belongs to the default run: profiles/rNN_bench_default.json)
$B >> $OUT 2>/dev/null
$B --op keyswitch >> $OUT 2>/dev/null
$B --op keyswitch --batch 4096 --steps 10 >> $OUT 2>/dev/null
$B --op automorphism >> $OUT 2>/dev/null
$B --op automorphism_add >> $OUT 2>/dev/null
$B --op trace --steps 5 >> $OUT 2>/dev/null
$B --op ggsw_expand_row >> $OUT 2>/dev/null
$B --op relinearize >> $OUT 2>/dev/null
$B --op relinearize --limbs 16 --batch 512 --steps 20 >> $OUT 2>/dev/null
$B --op keyswitch --limbs 16 --batch 512 --steps 20 >> $OUT 2>/dev/null
$B --op automorphism --limbs 16 --batch 512 --steps 20 >> $OUT 2>/dev/null
$B --op automorphism --limbs 16 --batch 512 --steps 20 --galois 1979 >> $OUT 2>/dev/null   # a Galois element without locality
$B --dsize 2 >> $OUT 2>/dev/null
$B --op keyswitch --dsize 2 >> $OUT 2>/dev/null
$B --base2k 14 >> $OUT 2>/dev/null
$B --n 4096 --limbs 4 --base2k 17 >> $OUT 2>/dev/null
$B --n 4096 --limbs 3 --base2k 18 --steps 100 >> $OUT 2>/dev/null   # poulpy-bench's default core shape (params.rs:113-121: n 2^12, base2k 18, k 54)
$B --n 4096 --limbs 3 --base2k 18 --op keyswitch --steps 100 >> $OUT 2>/dev/null
$B --n 4096 --limbs 3 --base2k 18 --op automorphism --steps 100 >> $OUT 2>/dev/null   # glwe_automorphism family on the two-kernel pipeline
$B --n 4096 --limbs 3 --base2k 18 --op automorphism_add --steps 100 >> $OUT 2>/dev/null
$B --n 2048 --limbs 4 --base2k 17 --op automorphism_add --steps 100 >> $OUT 2>/dev/null
$B --n 1024 --limbs 4 --base2k 17 --op automorphism_add --steps 100 >> $OUT 2>/dev/null
$B --n 4096 --limbs 4 --base2k 17 --op trace --steps 20 >> $OUT 2>/dev/null            # 12 / 10 steps of shift + automorphism_add_assign
$B --n 1024 --limbs 4 --base2k 17 --op trace --steps 20 >> $OUT 2>/dev/null
$B --n 131072 --batch 512 --steps 10 >> $OUT 2>/dev/null
$B --n 2048 --limbs 4 --base2k 17 --steps 100 >> $OUT 2>/dev/null
$B --n 1024 --limbs 4 --base2k 17 --steps 100 >> $OUT 2>/dev/null
$B --n 2048 --limbs 4 --base2k 17 --op keyswitch --steps 100 >> $OUT 2>/dev/null   # (N = 2048 key switch: the two-kernel pipeline; the external products above: k_small_one)
$B --n 1024 --limbs 4 --base2k 17 --op keyswitch --steps 100 >> $OUT 2>/dev/null
$B --n 8192 >> $OUT 2>/dev/null
$B --n 16384 >> $OUT 2>/dev/null
$B --n 32768 >> $OUT 2>/dev/null
$B --no-pin-key >> $OUT 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/bench_lines.jsonl"):
    try:
        d = json.loads(l)
    except Exception:
        continue
    r = d.get("roofline") or {}
    print(f'{d["value"]:12.0f} {d["unit"]:24s} {d["ms_per_step"]:8.3f} ms/step  parity={d.get("parity_sample", {}).get("ok")}  {d["config"]["workload"][:110]}  batch={d["config"]["batch_per_gpu"]}  {r.get("kernel_ms")}')
PY
python tools/bench_host_path.py 2>/dev/null; python tools/bench_host_path.py --pinned 2>/dev/null
# GLWE tensoring / multiplication at the configs[4] shape (tools/bench_tensor.py): profiles/rNN_tensor_lines.jsonl
T=gpurun_out/tensor_lines.jsonl; : > $T
python tools/bench_tensor.py 2>/dev/null | grep "^{" | tail -1 >> $T
python tools/bench_tensor.py --mode square 2>/dev/null | grep "^{" | tail -1 >> $T
python tools/bench_tensor.py --limbs 8 --batch 512 2>/dev/null | grep "^{" | tail -1 >> $T
python tools/bench_tensor.py --relin 2>/dev/null | grep "^{" | tail -1 >> $T
python tools/bench_tensor.py --relin --one-call 2>/dev/null | grep "^{" | tail -1 >> $T                  # the multiplication as ONE call, tensor in scratch (poulpy-ckks ckks_mul_into_default)
python tools/bench_tensor.py --relin --one-call --mode square 2>/dev/null | grep "^{" | tail -1 >> $T
python tools/bench_tensor.py --relin --one-call --limbs 8 --batch 512 2>/dev/null | grep "^{" | tail -1 >> $T
python -c "
import json
for l in open('gpurun_out/tensor_lines.jsonl'):
    d = json.loads(l); print('%12.0f %-22s parity=%s %s batch=%d %s' % (d['value'], d['unit'], d['parity_ok'], d['config']['workload'][:100], d['batch'], d['kernel_classes_launches_ms']))"
