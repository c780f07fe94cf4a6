#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
bash tools/pmc_decode_sq.sh r05_decode_sq 2>&1 | grep -A30 "parse_rows_kernel"
