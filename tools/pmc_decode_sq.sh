#!/bin/bash
# SQ issue-mix counters for the decode kernels (separate --pmc passes; kernel-trace only).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-decode_sq}; shift || true
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o $n -- python3 tools/decode_bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/$n.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run b SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
run c GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH
python3 - <<PY
import csv,glob,re
res={}
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(\w+_kernel)", r['Kernel_Name'])
        if 'v2p::' in r['Kernel_Name'] and m:
            res.setdefault(m.group(1),{}).setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
for k in sorted(res):
    print(k)
    for c,v in sorted(res[k].items()): print('   ',c, v[-1])
PY
