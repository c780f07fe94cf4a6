#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r05_call38; mkdir -p $OUT gpurun_out/profiles_r05
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cp profiles/traffic_latest.json gpurun_out/profiles_r05/ 2>/dev/null
SKIP_CEILING=1 bash tools/profile_round.sh r05_C5 --workload C5 --samples 10000 > gpurun_out/profile_C5.log 2>&1; tail -2 gpurun_out/profile_C5.log
