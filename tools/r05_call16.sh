#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 600 python tools/pad_probe.py C3 2000 2>&1 | tail -3
