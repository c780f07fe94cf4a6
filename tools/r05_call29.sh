#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call29; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_device_rows.py tests/test_gpu_batched.py -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for w in "C3 10000" "C4 2504"; do set -- $w
timeout 600 python tools/oneshot_bench.py --workload $1 --samples $2 --variants 0,22 --reps 3 > $OUT/ab_$1.json 2> $OUT/ab_$1.err; python3 - $OUT/ab_$1.json <<'P'
import json,sys
d=json.load(open(sys.argv[1])); print(d['workload'], d['summary'], d['steady_execute_ms'])
P
done
