#!/bin/bash
# round 4, item 2a: what would "eight ciphertexts per key fetch" buy at most?  Timing ablation of k_mid128r (-DPZ_MIDR_HALFKEY=1: every second
# key row of the product is never requested, results invalid) vs the product build, same box, alternating; then the stamps of both
for rep in 1 2; do
  bash tools/dbg/ab_libs.sh --args "--parity-samples 0" libpoulpy_hip.so variants/libpoulpy_hip_halfkey.so
done
echo "== key switch"
bash tools/dbg/ab_libs.sh --args "--op keyswitch --parity-samples 0" libpoulpy_hip.so variants/libpoulpy_hip_halfkey.so
