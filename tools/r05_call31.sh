#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call31; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_batched.py tests/test_gpu_whole_cohorts.py -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; python3 - <<'P'
import json
b=json.loads(open('gpurun_out/r05_call31/bench_default.json').read().strip().splitlines()[-1])
print(b['ms_per_step'], b['roofline']['frac'], b['roofline'].get('frac_of_box_fill'), {k:round(b['one_shot'][k],3) for k in ('total_ms','total_ms_gpu_busy_before','build_kernels_ms','first_execute_ms')}, b['host_packed'].get('ab_execute_ms_host_packed'), b['host_packed'].get('ab_execute_ms_device_built'), b['c2_cohort']['ms'])
P
