#!/bin/bash
# SQ issue counters of the device builder's parse kernel (rows_parse_kernel) on a C3 slice; run on the GPU box (gpurun).
# usage: tools/pmc_parse.sh [samples]   -> gpurun_out/pmc_parse/summary.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; N=${1:-2000}
OUT=$ROOT/gpurun_out/pmc_parse; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
CMD="python3 tools/build_bench.py --workload C3 --samples $N --reps 1 --no-exec --kernel 6"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/c -o c -- $CMD > $OUT/c.log 2>&1
python3 - > $OUT/summary.txt <<PY
import csv, glob
print("rows_parse_kernel<1, false, 0>, C3 $N samples, rocprofv3 --pmc (three passes), command: $CMD")
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    seen = {}
    for r in csv.DictReader(open(f)):
        if "rows_parse_kernel" in r["Kernel_Name"]:
            seen[r["Counter_Name"]] = seen.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, v in seen.items():
        print(f"{k:24s} {v:18,.0f}")
PY
cat $OUT/summary.txt; tail -1 $OUT/a.log | cut -c1-300
rm -rf $OUT/a $OUT/b $OUT/c
