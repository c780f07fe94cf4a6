#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call7
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_patch_image.py -q -x ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 300 python3 tools/oneshot_bench.py --workload C5 --samples 10000 --slices 1 --reps 3 --kernel 8 > $OUT/oneshot_C5_k8.json 2> $OUT/oneshot_C5_k8.err
python3 -c "import json;d=json.load(open('$OUT/oneshot_C5_k8.json'));print(json.dumps(d['summary']));print(d['runs']['two_calls_warm'])"
