#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call37; mkdir -p $OUT
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o $n -- python3 tools/pieces_probe.py C5 10000 > $OUT/$n.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run b SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
python3 - <<PY
import csv,glob,re,collections
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'stitch_pieces' in k or 'stitch_dense' in k:
            res['pieces' if 'pieces' in k else 'dense'][r['Counter_Name']].append(float(r['Counter_Value']))
for k in res:
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(res[k].items())})
PY
