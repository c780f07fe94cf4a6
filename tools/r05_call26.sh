#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call26; mkdir -p $OUT
for w in "C3 10000" "C4 2504"; do set -- $w
timeout 900 python tools/oneshot_bench.py --workload $1 --samples $2 --variants 0 --phase-mb 28,36,40,44,48,52,56 --reps 3 > $OUT/ph_$1.json 2> $OUT/ph_$1.err; python3 - $OUT/ph_$1.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(d['workload']); [print(k, v) for k,v in d['summary'].items() if 'warm' in k]
P
done
