#!/bin/bash
# Memory-side stall counters for the stitch kernel (separate --pmc passes; kernel-trace only).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-stall}; shift || true
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o $n -- python3 bench.py --no-cpu-baseline --no-pcie --no-device-build --no-verify --steps 2 --warmup 1 > $OUT/$n.log 2>&1; }
run a TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum
run b TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
run c TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run d TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run e TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum
python3 - <<PY
import csv,glob,json
res={}
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    seen={}
    for r in csv.DictReader(open(f)):
        if 'stitch' in r['Kernel_Name']:
            seen.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    for k,v in seen.items(): res[k]=v[-1]
json.dump(res, open("$OUT/stall_counters.json","w"), indent=1)
for k,v in sorted(res.items()): print(k, v)
PY
for n in a b c d e; do grep -il "error\|fail" $OUT/$n.log; done
