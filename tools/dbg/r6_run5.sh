#!/bin/bash
# round 6, GPU call 5: whole GPU suite at HEAD; the configs[3] lines with their checker (blind rotation + key switch, CPU sample); one-rank RCCL runs
# of the secondary benches (574 GGSWs through pz_bcast_key).
OUT=gpurun_out/r6_run5; mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $OUT/pytest_gpu.txt
BR=$OUT/bench_lines_br.jsonl; : > $BR
for sh in ref cbt n2048 n4096; do python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 2 2>$OUT/br_$sh.err | grep "^{" | tail -1 >> $BR; done
python tools/bench_blind_rotation.py --shape big --batch 1024 --cpu-cts 2 --reps 2 2>$OUT/br_big.err | grep "^{" | tail -1 >> $BR
for sh in cbt n2048 big; do python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 2 --reps 2 --with-keyswitch 2>$OUT/brks_$sh.err | grep "^{" | tail -1 >> $BR; done
python tools/bench_circuit_bootstrapping.py --batch 1024 --cpu-cts 1 2>$OUT/cbt1024.err | grep "^{" | tail -1 >> $BR
python tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 1 2>$OUT/cbt512.err | grep "^{" | tail -1 >> $BR
export POULPY_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
D=$OUT/one_rank_rccl.jsonl; : > $D
MASTER_PORT=29541 python tools/bench_blind_rotation.py --shape cbt --batch 1024 --cpu-cts 1 --gpus 1 --bcast cabi --with-keyswitch 2>$OUT/rccl_br.err | grep "^{" | tail -1 >> $D
MASTER_PORT=29542 python tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 1 --gpus 1 --bcast cabi 2>$OUT/rccl_cbt.err | grep "^{" | tail -1 >> $D
MASTER_PORT=29543 python tools/bench_tensor.py --relin --parity-samples 1 --gpus 1 --bcast cabi 2>$OUT/rccl_tensor.err | grep "^{" | tail -1 >> $D
unset POULPY_BENCH_FORCE_DIST RANK LOCAL_RANK WORLD_SIZE
python - <<'PY'
import json
for f in ("gpurun_out/r6_run5/bench_lines_br.jsonl", "gpurun_out/r6_run5/one_rank_rccl.jsonl"):
    print("==", f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: print("unparsable:", l[:200]); continue
        gb = d.get("gate_bootstrap") or {}
        print("%12.0f %-18s %-8s parity=%s cpu/s=%s margin=%.2g gpus=%s %s | gate bootstrap %s parity=%s" % (d["value"], d.get("unit", d["metric"])[:18], d.get("shape", ""), d.get("parity_on_cpu_sample", d.get("parity_ok")),
              ("%.3f" % d["cpu_port_1thread_per_s"]) if "cpu_port_1thread_per_s" in d else "-", d.get("rounding_margin") or 0, d.get("n_gpus"), d.get("parallelism", ""), ("%.0f/s" % gb["gate_bootstraps_per_s"]) if gb else "-", gb.get("parity_on_cpu_sample")))
PY
cat $OUT/pytest_gpu.txt; tail -2 $OUT/*.err | cut -c1-300 | tail -40
