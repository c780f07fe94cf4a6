#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call39; mkdir -p $OUT
SECONDS=0; timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; echo "bench wall ${SECONDS}s"
python3 - <<'P'
import json
b=json.loads(open('gpurun_out/r05_call39/bench_default.json').read().strip().splitlines()[-1])
print(b['metric'], b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline'].get('frac_of_box_fill'), b['roofline']['kernel'][:60])
print({k:round(b['one_shot'][k],3) for k in ('total_ms','total_ms_gpu_busy_before','build_kernels_ms','first_execute_ms')}, b['c2_cohort']['ms'], b['verified'])
P
