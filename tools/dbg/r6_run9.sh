#!/bin/bash
OUT=gpurun_out/r6_run9; mkdir -p $OUT
{
echo "== round-5 library, PyTorch's HIP runtime first (as pytest)"; POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_r5.so timeout 900 python tools/dbg/br_graph_repro.py plain torch 2>&1 | grep "torch\|br_big"
echo "== round-5 library, system runtime"; POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_r5.so timeout 900 python tools/dbg/br_graph_repro.py plain 2>&1 | grep "br_big"
echo "== HEAD (zero-fill kernels instead of memset nodes), PyTorch's HIP runtime first"; timeout 900 python tools/dbg/br_graph_repro.py plain torch 2>&1 | grep "torch\|br_big"
echo "== HEAD, structured tests under pytest"; timeout 900 python -m pytest tests/test_gpu_structured.py -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
} > $OUT/repro.txt 2>&1
cat $OUT/repro.txt | cut -c1-230
