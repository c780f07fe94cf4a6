#!/bin/bash
# HBM-side traffic counters for the stitch kernel (separate passes, as the guide prescribes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-mem}; shift || true
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -o p -- python3 bench.py --no-cpu-baseline --no-pcie --no-device-build --no-verify --steps 2 --warmup 1 "$@" > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    seen={}
    for r in csv.DictReader(open(f)):
        if 'stitch' in r['Kernel_Name']:
            seen.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    for k,v in seen.items(): print(k, v[-1])
PY
