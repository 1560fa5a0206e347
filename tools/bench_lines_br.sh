#!/bin/bash
# blind-rotation / gate-bootstrap / circuit-bootstrapping lines at HEAD (GPU box): one JSON line each into gpurun_out/bench_lines_br.jsonl, EVERY line with its
# checker (--cpu-cts: the oracle on two of the same ciphertexts, single-threaded: parity_on_cpu_sample + a CPU rate).  BASELINE configs[3] names "gate bootstrap
# (blind-rotate + keyswitch)": the --with-keyswitch lines (mod switch -> rotation -> lwe_from_glwe) for N = 1024 / 2048 / 2^14.
OUT=gpurun_out/bench_lines_br.jsonl; : > $OUT
for sh in ref cbt n2048 n4096; do python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 2 2>/dev/null | grep "^{" | tail -1 >> $OUT; done
python tools/bench_blind_rotation.py --shape big --batch 1024 --cpu-cts 2 --reps 2 2>/dev/null | grep "^{" | tail -1 >> $OUT
for sh in cbt n2048 big; do python tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 2 --reps 3 --with-keyswitch 2>/dev/null | grep "^{" | tail -1 >> $OUT; done
python tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 1 2>/dev/null | grep "^{" | tail -1 >> $OUT
python tools/bench_circuit_bootstrapping.py --batch 1024 --cpu-cts 1 2>/dev/null | grep "^{" | tail -1 >> $OUT
python - <<'PY'
import json
for l in open("gpurun_out/bench_lines_br.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    gb = d.get("gate_bootstrap") or {}
    print("%12.0f %-18s %-8s parity=%s cpu/s=%s margin=%.2g | gate bootstrap %s parity=%s" % (d["value"], d.get("unit", d["metric"])[:18], d.get("shape", ""), d.get("parity_on_cpu_sample"),
          ("%.3f" % d["cpu_port_1thread_per_s"]) if "cpu_port_1thread_per_s" in d else "-", d.get("rounding_margin") or 0, ("%.0f/s" % gb["gate_bootstraps_per_s"]) if gb else "-", gb.get("parity_on_cpu_sample")))
PY
