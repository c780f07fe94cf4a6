#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python3 tools/stage_probe.py C2 1000 0,64,96,128,192,256 2>&1 | tail -1
timeout 900 python3 tools/stage_probe.py C4 2504 0,20,24,28,32 2>&1 | tail -1
