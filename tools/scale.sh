#!/usr/bin/env bash
# 1 -> 8 GPU curve of the headline (north_star: "report external-products/sec and achieved HBM GB/s at 1, 2, 4 and 8 GPUs").
#   tools/scale.sh [GPU counts, default "1 2 4 8"] [-- extra bench.py arguments]
# Each run is `python bench.py --gpus N` (bench.py launches one rank per GPU itself); the first line's value is fed to the others as
# --ref-value, so every line carries scaling_efficiency = value / (N x the 1-GPU value).  Lines: gpurun_out/scale_lines.jsonl,
# table: stdout and gpurun_out/scale_table.txt.  Counts the node does not have are reported as "skipped" (bench.py exits 4).
set -u
cd "$(dirname "$0")/.."
counts=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do counts+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
[ ${#counts[@]} -eq 0 ] && counts=(1 2 4 8)
out=gpurun_out; mkdir -p "$out"; : > "$out/scale_lines.jsonl"
ref=0
for n in "${counts[@]}"; do
    args=(--gpus "$n" "$@")
    [ "$ref" != 0 ] && args+=(--ref-value "$ref")
    line=$(python bench.py "${args[@]}" 2> "$out/scale_n$n.err" | tail -n 1)
    rc=${PIPESTATUS[0]}
    if [ -z "$line" ] || ! printf '%s' "$line" | python -c 'import json,sys; json.loads(sys.stdin.read())' 2>/dev/null; then
        printf '{"n_gpus": %s, "skipped": true, "rc": %s}\n' "$n" "$rc" >> "$out/scale_lines.jsonl"
        continue
    fi
    printf '%s\n' "$line" >> "$out/scale_lines.jsonl"
    [ "$ref" = 0 ] && [ "$n" = 1 ] && ref=$(printf '%s' "$line" | python -c 'import json,sys; print(json.loads(sys.stdin.read())["value"])')
done
python tools/scale_table.py "$out/scale_lines.jsonl" | tee "$out/scale_table.txt"
