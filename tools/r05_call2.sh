#!/bin/bash
# round 5, GPU call 2: the one-call path -- parity tests, then the slice sweep on C3 whole and C2
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_device_rows.py tests/test_gpu_device_build.py -x -q --durations=8 ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 900 python3 tools/oneshot_bench.py --workload C3 --samples 10000 --slices 1,2,4,6,8,12,16,24 --reps 3 --oracle > $OUT/oneshot_C3.json 2> $OUT/oneshot_C3.err
tail -c 400 $OUT/oneshot_C3.err
python3 -c "import json;d=json.load(open('$OUT/oneshot_C3.json'));print(json.dumps(d['summary'],indent=0));print(d['steady_execute_ms_sliced_image'],d['steady_execute_ms_one_piece_image'])"
timeout 600 python3 tools/oneshot_bench.py --workload C2 --samples 1000 --slices 1,2,4,6,8 --reps 3 > $OUT/oneshot_C2.json 2> $OUT/oneshot_C2.err
tail -c 400 $OUT/oneshot_C2.err
python3 -c "import json;d=json.load(open('$OUT/oneshot_C2.json'));print(json.dumps(d['summary'],indent=0));print(d['steady_execute_ms_sliced_image'],d['steady_execute_ms_one_piece_image'])"
