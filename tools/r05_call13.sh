#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call13
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_device_rows.py -q -x > $OUT/pytest.log 2>&1; head -60 $OUT/pytest.log
