#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
bash tools/profile_decode.sh > gpurun_out/profile_decode.log 2>&1; tail -5 gpurun_out/profile_decode.log
timeout 600 python tools/decode_bench.py --format rich > gpurun_out/prof_decode/bench_rich.json 2>/dev/null; tail -c 300 gpurun_out/prof_decode/bench_rich.json
bash tools/pmc_decode_sq.sh r05_decode_sq > gpurun_out/r05_sq_counters_decode.txt 2>&1; grep -A24 "parse_rows_kernel" gpurun_out/r05_sq_counters_decode.txt | head -26
