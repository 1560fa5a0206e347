#!/bin/bash
# round 6, GPU call 33: one-rank RCCL runs at HEAD: bench.py, the one-call multiplication (tensor key through pz_bcast_key), the blind rotation with its 16-bit accumulator
OUT=gpurun_out/r6_run33; mkdir -p $OUT
export POULPY_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
D=$OUT/one_rank_rccl.jsonl; : > $D
MASTER_PORT=29551 python tools/bench_tensor.py --relin --one-call --parity-samples 1 --gpus 1 --bcast cabi 2>$OUT/rccl_tensor.err | grep "^{" | tail -1 >> $D
MASTER_PORT=29552 python tools/bench_blind_rotation.py --shape big --batch 1024 --cpu-cts 1 --reps 2 --gpus 1 --bcast cabi --with-keyswitch 2>$OUT/rccl_br.err | grep "^{" | tail -1 >> $D
MASTER_PORT=29553 python bench.py --gpus 1 --bcast cabi --no-cpu-baseline --sustained-seconds 0 --op automorphism --limbs 16 --batch 512 --steps 20 --parity-samples 2 2>$OUT/rccl_bench.err | grep "^{" | tail -1 >> $D
unset POULPY_BENCH_FORCE_DIST RANK LOCAL_RANK WORLD_SIZE
python - <<'PY'
import json
for l in open("gpurun_out/r6_run33/one_rank_rccl.jsonl"):
    d = json.loads(l)
    print("%10.0f %-22s n_gpus=%s rccl_ranks=%s parallelism=%s parity=%s" % (d["value"], d.get("unit", "")[:22], d.get("n_gpus"), d.get("rccl_ranks", (d.get("config") or {}).get("rccl_ranks")), str(d.get("parallelism", (d.get("config") or {}).get("parallelism")))[:60], d.get("parity_ok", d.get("parity_on_cpu_sample", (d.get("parity_sample") or {}).get("ok")))))
PY
