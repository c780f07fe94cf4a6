#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call10
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_patch_image.py -q -x 2>&1 | tail -3
timeout 300 python3 tools/image_kinds.py --workload C5 --samples 10000 --kernels 7 8 2>/dev/null | cut -c1-700
timeout 300 python3 tools/image_kinds.py --workload C5 --samples 10000 --kernels 7 8 --mix 1 0 0 0 0 0 2>/dev/null | cut -c1-700
