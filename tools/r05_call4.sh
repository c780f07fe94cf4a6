#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call4
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 600 python3 tools/phase_sweep.py --workload C3 --samples 10000 --launch-forms --phases 20 28 40 56 --rounds 7 > $OUT/dual_C3.json 2> $OUT/dual_C3.err
cat $OUT/dual_C3.json; tail -c 300 $OUT/dual_C3.err
timeout 600 python3 tools/phase_sweep.py --workload C2 --samples 1000 --launch-forms --phases 64 128 --rounds 7 > $OUT/dual_C2.json 2> $OUT/dual_C2.err
cat $OUT/dual_C2.json
timeout 600 python3 tools/phase_sweep.py --workload C4 --samples 2504 --launch-forms --phases 28 56 --rounds 5 > $OUT/dual_C4.json 2> $OUT/dual_C4.err
cat $OUT/dual_C4.json
