#!/bin/bash
# SQ counter passes + kernel statistics for the blind-rotation kernels (run on the GPU box through gpurun): k_br_fused (ref, cbt shapes),
# k_br_block_lds<2,6,3> + k_small_inv (n2048), k_br_block_lds<2,9,3> (circuit bootstrapping), k_mid128<BR> + k_inv_tail (n4096).
# Counter passes carry --kernel-trace only (no --stats / trace domains with --pmc on this pool).  Summaries: gpurun_out/prof_br_pmc/*.txt
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_br_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASS_A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
PASS_B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"
run() {   # tag, program args...
  tag=$1; shift
  rocprofv3 --pmc $PASS_A --kernel-trace --output-format csv -d $OUT/${tag}_a -- python3 "$@" > $OUT/${tag}_a.log 2>&1
  rocprofv3 --pmc $PASS_B --kernel-trace --output-format csv -d $OUT/${tag}_b -- python3 "$@" > $OUT/${tag}_b.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_s -- python3 "$@" > $OUT/${tag}_s.log 2>&1
}
for sh in ${BR_PMC_SHAPES:-ref cbt n2048 n4096}; do
  run $sh $REPO/tools/bench_blind_rotation.py --shape $sh --batch 1024 --cpu-cts 0 --reps 1
done
[ -n "$BR_PMC_NO_CBT" ] || run circuit $REPO/tools/bench_circuit_bootstrapping.py --batch 512 --cpu-cts 0 --reps 1
cd $OUT
python3 - <<'PY'
import csv, glob, collections, os, re
def short(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*", "", k)
    return k.replace("pz::", "")[:90]
for tagdir in sorted(set(d[:-2] for d in glob.glob("*_a"))):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for kind in ("a", "b"):
        for f in glob.glob(f"{tagdir}_{kind}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = (short(row.get("Kernel_Name", "?")), row.get("Counter_Name"))
                agg[k][0] += 1
                agg[k][1] += float(row.get("Counter_Value", 0))
            os.remove(f)
    kernels = sorted(set(k for k, _ in agg))
    with open(f"{tagdir}_pmc.txt", "w") as o:
        for k in kernels:
            if not (k.startswith("k_") or "k_" in k[:6]):
                continue
            c = {cn: agg[(kk, cn)] for (kk, cn) in agg if kk == k}
            n = max(v[0] for v in c.values())
            per = {cn: v[1] / max(v[0], 1) for cn, v in c.items()}
            o.write(f"== {k}   dispatches per pass: {n}\n")
            for cn in sorted(per):
                o.write(f"   {cn:24s} {per[cn]:16.4g}\n")
            wc = per.get("SQ_WAVE_CYCLES")
            if wc:
                o.write("   -- shares of SQ_WAVE_CYCLES: " + "  ".join(f"{cn[3:]} {per[cn] / wc:.3f}" for cn in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if cn in per) + "\n")
            if per.get("SQ_INSTS_VALU") and per.get("SQ_ACTIVE_INST_VALU"):
                o.write(f"   -- quad-cycles per VALU instruction: {per['SQ_ACTIVE_INST_VALU'] / per['SQ_INSTS_VALU']:.3f}   LDS bank-conflict cycles per LDS instruction: {per.get('SQ_LDS_BANK_CONFLICT', 0) / max(per.get('SQ_INSTS_LDS', 1), 1):.3f}\n")
    for f in glob.glob(f"{tagdir}_s/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(f"{tagdir}_stats.txt", "w") as o:
            for r in rows[:12]:
                o.write(f"{short(r.get('Name', '?')):90s} calls {r.get('Calls'):>5s} total_ns {r.get('TotalDurationNs'):>12s} avg_ns {r.get('AverageNs'):>12s} pct {r.get('Percentage')}\n")
    for f in glob.glob(f"{tagdir}_s/**/*kernel_trace.csv", recursive=True):
        os.remove(f)
    for f in glob.glob(f"{tagdir}_[ab]/**/*kernel_trace.csv", recursive=True):
        os.remove(f)
PY
ls $OUT | head -40
