#!/bin/bash
OUT=gpurun_out/r6_run8; mkdir -p $OUT
{
echo "== full order, graphs on"; timeout 900 python tools/dbg/br_graph_repro.py plain 2>&1 | grep -v "^$" | tail -60
echo "== full order, graphs off on the N=2^14 module"; timeout 900 python tools/dbg/br_graph_repro.py nographs 2>&1 | grep "br_big"
} > $OUT/repro.txt 2>&1
cat $OUT/repro.txt | cut -c1-250
