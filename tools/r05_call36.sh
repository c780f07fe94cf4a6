#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pieces.py -q -x 2>&1 | tail -2
timeout 600 python3 tools/pieces_probe.py C5 10000 0 0,16,32,64,128,256 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms'])"
