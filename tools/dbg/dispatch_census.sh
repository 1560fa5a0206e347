#!/bin/bash
# census of the kernel instantiations the GPU suite and the bench shapes dispatch (experiment build: POULPY_BUILD_DEFS=-DPZ_EXPERIMENT
# POULPY_BUILD_TAG=census): every new dispatch note of a module is appended to gpurun_out/dispatch_census_raw.txt; the sorted set with
# counts goes to gpurun_out/dispatch_census.txt
export POULPY_HIP_LIB=$PWD/poulpy_amd/variants/libpoulpy_hip_census.so
export POULPY_DBG_DISPATCH_LOG=$PWD/gpurun_out/dispatch_census_raw.txt
: > $POULPY_DBG_DISPATCH_LOG
python -m pytest tests -m gpu -q -x -k "not canary and not launcher and not cpp_abi" 2>&1 | tail -2
echo "---- bench shapes" >> $POULPY_DBG_DISPATCH_LOG
for sh in ref cbt n2048 n4096; do python tools/bench_blind_rotation.py --shape $sh --cpu-cts 0 --reps 1 > /dev/null 2>&1; done
python tools/bench_blind_rotation.py --shape big --batch 64 --cpu-cts 0 --reps 1 > /dev/null 2>&1
python tools/bench_blind_rotation.py --shape cbt --block-size 1 --n-lwe 64 --cpu-cts 0 --reps 1 > /dev/null 2>&1
python tools/bench_circuit_bootstrapping.py --batch 64 --cpu-cts 0 --reps 1 > /dev/null 2>&1
sed 's/ lds=[0-9]*//' $POULPY_DBG_DISPATCH_LOG | sort | uniq -c | sort -k2 > gpurun_out/dispatch_census.txt
wc -l gpurun_out/dispatch_census.txt
