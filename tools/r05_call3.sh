#!/bin/bash
# round 5, GPU call 3: one slice by default + compaction beside the cutter + the row map with the rest; sharded host mode; launcher forms
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_device_rows.py tests/test_gpu_harness.py tests/test_gpu_device_build_fuzz.py -x -q --durations=8 ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 900 python3 tools/oneshot_bench.py --workload C3 --samples 10000 --slices 1,2 --reps 5 > $OUT/oneshot_C3.json 2> $OUT/oneshot_C3.err
tail -c 400 $OUT/oneshot_C3.err
python3 -c "import json;d=json.load(open('$OUT/oneshot_C3.json'));print(json.dumps(d['summary'],indent=0));print(d['steady_execute_ms_sliced_image'],d['steady_execute_ms_one_piece_image']);print(d['runs']['one_call_S1_warm'])"
timeout 600 python3 tools/oneshot_bench.py --workload C2 --samples 1000 --slices 1 --reps 5 > $OUT/oneshot_C2.json 2> $OUT/oneshot_C2.err
python3 -c "import json;d=json.load(open('$OUT/oneshot_C2.json'));print(json.dumps(d['summary'],indent=0));print(d['runs']['one_call_S1_warm'])"
timeout 600 python3 tools/oneshot_bench.py --workload C5 --samples 10000 --slices 1 --reps 3 > $OUT/oneshot_C5.json 2> $OUT/oneshot_C5.err
python3 -c "import json;d=json.load(open('$OUT/oneshot_C5.json'));print(json.dumps(d['summary'],indent=0));print(d['runs']['one_call_S1_warm'])"
timeout 600 python3 tools/phase_sweep.py --workload C3 --samples 10000 --launch-forms --phases 28 64 --rounds 5 > $OUT/phase_forms_C3.json 2> $OUT/phase_forms_C3.err
cat $OUT/phase_forms_C3.json
timeout 600 python3 tools/phase_sweep.py --workload C2 --samples 1000 --launch-forms --phases 64 --rounds 5 > $OUT/phase_forms_C2.json 2> $OUT/phase_forms_C2.err
cat $OUT/phase_forms_C2.json
timeout 300 python3 tools/first_execute.py --execs 6 > $OUT/first_execute.json 2> $OUT/first_execute.err
python3 -c "import json;d=json.load(open('$OUT/first_execute.json'));[print(k,v.get('sclk_start'),v.get('sclk_after_build'),v.get('sclk_end')) for k,v in d['scenarios'].items()]"
