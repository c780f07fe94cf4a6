#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=gpurun_out/r05_call35; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pieces.py -q -x > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
