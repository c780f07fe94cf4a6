#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python3 tools/stage_probe.py C3 1250 0,28,44 23,25 2>&1 | tail -1
timeout 900 python3 tools/stage_probe.py C3 2500 0,44 23,25 2>&1 | tail -1
