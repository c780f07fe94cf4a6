#!/bin/bash
# SQ issue counters of the device builder's parse kernel (rows_parse_kernel) inside the ONE call; run on the GPU box (gpurun).
# usage: tools/pmc_parse.sh [workload] [samples] [kernel]   -> gpurun_out/pmc_parse_<workload>/summary.txt
#   C3 2000 0 -> rows_parse_kernel<1, false, 0> (wave image);  C5 10000 0 -> <3, false, 0> (tile image);  C5 10000 7 -> <2, false, 0> (dense rows image)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-C3}; N=${2:-2000}; K=${3:-0}
OUT=$ROOT/gpurun_out/pmc_parse_${W}_k$K; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
CMD="python3 tools/oneshot_once.py --workload $W --samples $N --kernel $K"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/c -o c -- $CMD > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- $CMD > $OUT/t.log 2>&1
python3 - > $OUT/summary.txt <<PY
import csv, glob
print("rows_parse_kernel inside ONE v2p_batch_build_and_execute, $W $N samples, kernel argument $K; rocprofv3 --pmc (three passes) + --kernel-trace --stats; command: $CMD")
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    seen, name = {}, ""
    for r in csv.DictReader(open(f)):
        if "rows_parse_kernel" in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            seen[r["Counter_Name"]] = seen.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, v in seen.items():
        print(f"{k:24s} {v:18,.0f}   {name[:60]}")
for f in glob.glob("$OUT/t/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("stats:", r["Name"][:70], "calls", r["Calls"], "avg_ns", r["AverageNs"], "pct", r["Percentage"])
PY
cat $OUT/summary.txt; tail -1 $OUT/a.log | cut -c1-400
rm -rf $OUT/a $OUT/b $OUT/c $OUT/t
