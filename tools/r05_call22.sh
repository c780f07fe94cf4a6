#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
bash tools/r05_call20.sh
bash tools/pmc_decode_sq.sh r05_decode_sq 2>&1 | grep -A24 "parse_rows_kernel" | grep "INSTS_VALU\|INSTS_SALU\|GUI_ACTIVE\|INSTS_LDS\|WAIT_INST_ANY\|WAVE_CYCLES"
