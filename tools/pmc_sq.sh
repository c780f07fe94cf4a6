#!/bin/bash
# SQ issue-mix counters for the stitch kernel (one pass, 8 SQ slots)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-sq}; shift || true
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -o a -- python3 bench.py --no-cpu-baseline --no-pcie --no-device-build --no-verify --steps 2 --warmup 1 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/b -o b -- python3 bench.py --no-cpu-baseline --no-pcie --no-device-build --no-verify --steps 2 --warmup 1 "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/c -o c -- python3 bench.py --no-cpu-baseline --no-pcie --no-device-build --no-verify --steps 2 --warmup 1 "$@" > $OUT/c.log 2>&1
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    seen={}
    for r in csv.DictReader(open(f)):
        if 'stitch' in r['Kernel_Name']:
            seen.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    for k,v in seen.items(): print(k, v[-1])
PY
tail -2 $OUT/c.log | head -1 | cut -c1-300
