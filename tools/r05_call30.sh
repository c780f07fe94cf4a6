#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 900 python3 tools/stage_probe.py C3 10000 0,28,44 2>&1 | tail -1
