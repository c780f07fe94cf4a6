#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call11
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_oneshot.py -q -x -k degenerate 2>&1 | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
