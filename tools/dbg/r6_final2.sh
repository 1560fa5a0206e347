#!/bin/bash
# round 6, final measurement, part 2: the rocprofv3 passes of the headline (tools/prof.sh, now without the sustained leg) and of the one-call multiplication
R=06; O=gpurun_out
bash tools/prof.sh > $O/r${R}_prof.log 2>&1
cp $O/prof/trace_steady.txt $O/r${R}_kernel_steady.txt 2>/dev/null
find $O/prof/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r${R}_kernel_stats.csv
cp $O/prof/pmc_fetch_summary.txt $O/r${R}_pmc_fetch_summary.txt; cp $O/prof/pmc_write_summary.txt $O/r${R}_pmc_write_summary.txt
python tools/traffic_json.py $O/r${R}_pmc_fetch_summary.txt $O/r${R}_pmc_write_summary.txt $O/r${R}_traffic.json 1024 > $O/r${R}_traffic.txt 2>&1
cat $O/r${R}_kernel_steady.txt; cat $O/r${R}_traffic.txt
bash tools/prof_tensor.sh --relin --one-call > $O/r${R}_prof_tensor_onecall.log 2>&1
mkdir -p $O/r6_final2; cp $O/prof_tensor/*.txt $O/r6_final2/
cat $O/r6_final2/kernel_stats.txt | cut -c1-150
ls $O/.graft_exec_refused 2>/dev/null && tail -2 $O/.graft_exec_refused
