#!/bin/bash
# round 6, GPU call 3: whole GPU suite at HEAD, tensoring A/B (round-5 library vs HEAD), k_mid_cnv3 stamps, rocprofv3 evidence of the tensoring,
# the configs[3] lines with their checker (blind rotation + key switch, CPU sample), one-rank RCCL runs of the secondary benches.
OUT=gpurun_out/r6_run3; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $OUT/pytest_gpu.txt
{
echo "== A/B"
for rep in 1 2; do
for lib in variants/libpoulpy_hip_r5.so libpoulpy_hip.so; do
  export POULPY_HIP_LIB=$PWD/poulpy_amd/$lib
  for args in "" "--relin" "--mode square" "--limbs 8 --batch 512"; do
    python tools/bench_tensor.py --parity-samples 1 $args 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-36s %-22s %8.0f %s parity=%s margin=%.2g %s' % ('$lib', '$args', d['value'], d['unit'], d['parity_ok'], d['rounding_margin'] or 0, d['kernel_classes_launches_ms']))"
  done
done
done
unset POULPY_HIP_LIB
echo "== stamps: HEAD, apply"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp1.so | head -12
echo "== stamps: HEAD, square"
tools/dbg/cnv_stamps.sh variants/libpoulpy_hip_cnvstamp1.so --mode square | head -12
} > $OUT/ab.txt 2>&1
bash tools/prof_tensor.sh > $OUT/prof_tensor.log 2>&1
mkdir -p $OUT/prof_tensor && cp gpurun_out/prof_tensor/*.txt $OUT/prof_tensor/ 2>/dev/null
# configs[3] with the checker on every line
BR=$OUT/bench_lines_br.jsonl; : > $BR
for sh in ref cbt n2048 n4096; do python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 2 2>/dev/null | grep "^{" | tail -1 >> $BR; done
python tools/bench_blind_rotation.py --shape big --batch 1024 --cpu-cts 2 --reps 2 2>/dev/null | grep "^{" | tail -1 >> $BR
for sh in cbt n2048 big; do python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 2 --reps 2 --with-keyswitch 2>/dev/null | grep "^{" | tail -1 >> $BR; done
python tools/bench_circuit_bootstrapping.py --batch 1024 --cpu-cts 1 2>/dev/null | grep "^{" | tail -1 >> $BR
python tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 1 2>/dev/null | grep "^{" | tail -1 >> $BR
# one-rank RCCL: the keys of the secondary benches through pz_bcast_key (574 GGSWs in one tensor for the rotation)
export POULPY_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
D=$OUT/one_rank_rccl.jsonl; : > $D
MASTER_PORT=29541 python tools/bench_blind_rotation.py --shape cbt --batch 1024 --cpu-cts 1 --gpus 1 --bcast cabi --with-keyswitch 2>$OUT/rccl_br.err | grep "^{" | tail -1 >> $D
MASTER_PORT=29542 python tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 1 --gpus 1 --bcast cabi 2>$OUT/rccl_cbt.err | grep "^{" | tail -1 >> $D
MASTER_PORT=29543 python tools/bench_tensor.py --relin --parity-samples 1 --gpus 1 --bcast cabi 2>$OUT/rccl_tensor.err | grep "^{" | tail -1 >> $D
unset POULPY_BENCH_FORCE_DIST RANK LOCAL_RANK WORLD_SIZE
python - <<'PY'
import json
for f in ("gpurun_out/r6_run3/bench_lines_br.jsonl", "gpurun_out/r6_run3/one_rank_rccl.jsonl"):
    print("==", f)
    for l in open(f):
        try: d = json.loads(l)
        except Exception: print("unparsable:", l[:200]); continue
        gb = d.get("gate_bootstrap") or {}
        print("%12.0f %-18s %-8s parity=%s cpu/s=%s margin=%.2g gpus=%s %s | gate bootstrap %s parity=%s" % (d["value"], d.get("unit", d["metric"])[:18], d.get("shape", ""), d.get("parity_on_cpu_sample", d.get("parity_ok")),
              ("%.2f" % d["cpu_port_1thread_per_s"]) if "cpu_port_1thread_per_s" in d else "-", d.get("rounding_margin") or 0, d.get("n_gpus"), d.get("parallelism", ""), ("%.0f/s" % gb["gate_bootstraps_per_s"]) if gb else "-", gb.get("parity_on_cpu_sample")))
PY
cat $OUT/pytest_gpu.txt; tail -30 $OUT/ab.txt | cut -c1-220
