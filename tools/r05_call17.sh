#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call17; mkdir -p $OUT
for v in 0 22; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_v$v -o p -- python3 tools/pad_probe.py C3 2000 $v > $OUT/log_v$v.txt 2>&1
  grep exec5 $OUT/log_v$v.txt
  f=$(find $OUT/prof_v$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'P'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:7]:
    print(r[0][:60].ljust(60), r[1], r[3], r[5], r[6])
P
  for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $pmc | tr ' ' '_')
    timeout 600 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_v${v}_$tag -o c -- python3 tools/pad_probe.py C3 2000 $v > $OUT/log_pmc_v${v}_$tag.txt 2>&1
    f=$(find $OUT/pmc_v${v}_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" <<'P'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
rd=csv.DictReader(open(sys.argv[1]))
for r in rd:
    k=r['Kernel_Name'][:40]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
for k in agg:
    if 'stitchw' in k or 'touch' in k:
        print(k, {c:(round(v/cnt[(k,c)],1), cnt[(k,c)]) for c,v in agg[k].items()})
P
  done
done
