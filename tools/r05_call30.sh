#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python3 tools/stage_probe.py C3 1250 0,20,28,36,44,56,64,96 23 2>&1 | tail -1
timeout 900 python3 tools/stage_probe.py C3 2500 0,28,36,44,56,64 23 2>&1 | tail -1
timeout 900 python3 tools/stage_probe.py C3 5000 0,28,36,44,56 23,25 2>&1 | tail -1
