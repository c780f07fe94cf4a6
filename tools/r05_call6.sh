#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call6
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_patch_image.py -q -x --durations=5 ) > $OUT/pytest.log 2>&1
tail -40 $OUT/pytest.log
timeout 600 python3 tools/oneshot_bench.py --workload C5 --samples 10000 --slices 1 --reps 3 --kernel 8 > $OUT/oneshot_C5_k8.json 2> $OUT/oneshot_C5_k8.err
tail -c 600 $OUT/oneshot_C5_k8.err
python3 -c "import json;d=json.load(open('$OUT/oneshot_C5_k8.json'));print(json.dumps(d['summary'],indent=0));print(d['steady_execute_ms_sliced_image'],d['steady_execute_ms_one_piece_image'], d['descriptors'], d['chunks']);print(d['runs']['two_calls_warm'])"
