#!/usr/bin/env bash
# 1 -> 8 GPU curve of the headline (north_star: "report external-products/sec and achieved HBM GB/s at 1, 2, 4 and 8 GPUs").
#   tools/scale.sh [--bench headline|br|cbt|tensor] [GPU counts, default "1 2 4 8"] [-- extra bench arguments]
# --bench br / cbt / tensor: the same curve for tools/bench_blind_rotation.py / bench_circuit_bootstrapping.py / bench_tensor.py (BASELINE configs[3], [4])
# Each run is `python bench.py --gpus N` (bench.py launches one rank per GPU itself); the first line's value is fed to the others as
# --ref-value, so every line carries scaling_efficiency = value / (N x the 1-GPU value).  Lines: gpurun_out/scale_lines.jsonl,
# table: stdout and gpurun_out/scale_table.txt.  Counts the node does not have are reported as "skipped" (bench.py exits 4).
set -u
cd "$(dirname "$0")/.."
bench=bench.py
if [ "${1:-}" = "--bench" ]; then
    case "$2" in
        headline) bench=bench.py ;;
        br) bench=tools/bench_blind_rotation.py ;;
        cbt) bench=tools/bench_circuit_bootstrapping.py ;;
        tensor) bench=tools/bench_tensor.py ;;
        *) echo "unknown --bench $2" >&2; exit 2 ;;
    esac
    shift 2
fi
counts=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do counts+=("$1"); shift; done
[ "${1:-}" = "--" ] && shift
[ ${#counts[@]} -eq 0 ] && counts=(1 2 4 8)
out=gpurun_out; mkdir -p "$out"; : > "$out/scale_lines.jsonl"
ref=0
for n in "${counts[@]}"; do
    args=(--gpus "$n" "$@")
    [ "$ref" != 0 ] && args+=(--ref-value "$ref")
    # bench.py's own exit status (not tail's): 4 = fewer devices than ranks ("skipped"); anything else non-zero is a FAILED run
    python $bench "${args[@]}" > "$out/scale_n$n.out" 2> "$out/scale_n$n.err"
    rc=$?
    line=$(tail -n 1 "$out/scale_n$n.out")
    if [ "$rc" != 0 ] || [ -z "$line" ] || ! printf '%s' "$line" | python -c 'import json,sys; json.loads(sys.stdin.read())' 2>/dev/null; then
        if [ "$rc" = 4 ]; then printf '{"n_gpus": %s, "skipped": true, "rc": %s}\n' "$n" "$rc" >> "$out/scale_lines.jsonl"
        else printf '{"n_gpus": %s, "failed": true, "rc": %s}\n' "$n" "$rc" >> "$out/scale_lines.jsonl"; fi
        continue
    fi
    printf '%s\n' "$line" >> "$out/scale_lines.jsonl"
    [ "$ref" = 0 ] && [ "$n" = 1 ] && ref=$(printf '%s' "$line" | python -c 'import json,sys; print(json.loads(sys.stdin.read())["value"])')
done
python tools/scale_table.py "$out/scale_lines.jsonl" | tee "$out/scale_table.txt"
