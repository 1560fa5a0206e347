#!/bin/bash
# what the driver runs at round end, on the final build: the whole GPU suite, smoke(), the default bench line
OUT=gpurun_out/r6_validate; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/pytest_gpu.txt; tail -2 $OUT/smoke.txt; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_validate/bench_default.json"))
print(d["value"], d["ms_per_step"], d["sustained"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["parity_sample"]["ok"], d["rounding_margin"])
PY
