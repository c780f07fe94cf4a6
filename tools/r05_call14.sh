#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call14
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_device_rows.py -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 600 python tools/oneshot_bench.py --workload C3 --samples 10000 --variants 0,22,0,22 --reps 3 > $OUT/ab_C3.json 2> $OUT/ab_C3.err; tail -c 400 $OUT/ab_C3.json
timeout 300 python tools/oneshot_bench.py --workload C2 --variants 0,22 --reps 3 > $OUT/ab_C2.json 2> $OUT/ab_C2.err; tail -c 300 $OUT/ab_C2.json
timeout 300 python tools/oneshot_bench.py --workload C4 --samples 2504 --variants 0,22 --reps 3 > $OUT/ab_C4.json 2> $OUT/ab_C4.err; tail -c 300 $OUT/ab_C4.json
