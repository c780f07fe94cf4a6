#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles_r05
cp profiles/traffic_latest.json gpurun_out/profiles_r05/ 2>/dev/null
bash tools/profile_round.sh r05_C3whole > gpurun_out/profile_C3whole.log 2>&1; tail -3 gpurun_out/profile_C3whole.log
SKIP_CEILING=1 bash tools/profile_round.sh r05_C4whole --workload C4 --samples 2504 > gpurun_out/profile_C4whole.log 2>&1; tail -2 gpurun_out/profile_C4whole.log
SKIP_CEILING=1 bash tools/profile_round.sh r05_C5 --workload C5 --samples 10000 > gpurun_out/profile_C5.log 2>&1; tail -2 gpurun_out/profile_C5.log
SKIP_CEILING=1 bash tools/profile_round.sh r05_C2 --workload C2 --samples 1000 > gpurun_out/profile_C2.log 2>&1; tail -2 gpurun_out/profile_C2.log
timeout 1200 python bench.py > gpurun_out/profiles_r05/r05_bench_default_line.json 2> gpurun_out/profiles_r05/bench_default.err; echo "bench rc=$?"
