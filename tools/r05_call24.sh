#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for o in 0,22 22,0 0 22; do echo "order $o"; timeout 600 python3 tools/pad_probe.py C3 10000 $o 2>&1 | tail -1; done
