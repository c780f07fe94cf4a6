#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call15
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python tools/oneshot_bench.py --workload C3 --samples 10000 --variants 0,22 --reps 2 > $OUT/ab_C3.json 2> $OUT/ab_C3.err; python - <<'P'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05_call15/ab_C3.json'))
print(d['summary'], d['steady_execute_ms'])
P
