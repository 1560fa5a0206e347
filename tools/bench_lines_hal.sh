#!/bin/bash
# per-op HAL device lines at HEAD (GPU box): gpurun_out/bench_lines_hal.jsonl + a table
OUT=gpurun_out/bench_lines_hal.jsonl; : > $OUT
H="python tools/bench_hal_ops.py"
# the metric shape (N = 2^16, 8 limbs, 2 columns, 1024 containers)
for op in dft idft svp normalize; do $H --op $op >> $OUT 2>/dev/null; done
$H --op vmp >> $OUT 2>/dev/null
# FFT sweep of the reference (benches/fft.rs:21: m = 2^9 .. 2^15), 1 GiB of polynomials per call
for ln in 10 11 12 13 14 15 16; do
  $H --op dft --n $((1 << ln)) --cols 1 --limbs 1 --batch $(( (1 << 27) >> ln )) >> $OUT 2>/dev/null
  $H --op idft --n $((1 << ln)) --cols 1 --limbs 1 --batch $(( (1 << 27) >> ln )) >> $OUT 2>/dev/null
done
# VMP sweep of the reference (src/params.rs:72-84: [log_n, rows, cols_in, cols_out, size]), 256 vectors per call
for s in "10 2 1 2 3" "11 4 1 2 5" "12 7 1 2 8" "13 15 1 2 16" "14 31 1 2 32"; do
  set -- $s
  $H --op vmp --n $((1 << $1)) --rows $2 --cols-in $3 --cols-out $4 --limbs $5 --batch 256 >> $OUT 2>/dev/null
done
python - <<'PY'
import json
for l in open("gpurun_out/bench_lines_hal.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]
    print("%14.0f %-26s %8.3f ms/step  %7.0f GB/s  frac %.3f  %s  %s" % (d["value"], d["unit"], d["ms_per_step"], r["achieved"], r["frac"],
          ("fp64 %.1f TF" % r["fp64_tflops"]) if "fp64_tflops" in r else "", d["config"]["workload"][:110]))
PY
