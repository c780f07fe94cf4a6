#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call8
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 300 python3 tools/image_kinds.py --workload C5 --samples 10000 --kernels 7 8 > $OUT/kinds_C5.json 2> $OUT/err1; cat $OUT/kinds_C5.json; tail -c 300 $OUT/err1
timeout 300 python3 tools/image_kinds.py --workload C5 --samples 10000 --kernels 7 8 --mix 1 0 0 0 0 0 > $OUT/kinds_C5_missense.json 2> $OUT/err2; cat $OUT/kinds_C5_missense.json; tail -c 300 $OUT/err2
timeout 300 python3 tools/image_kinds.py --workload C5 --samples 10000 --kernels 7 8 --mix 0.8 0.1 0.1 0 0 0 > $OUT/kinds_C5_mix811.json 2> $OUT/err3; cat $OUT/kinds_C5_mix811.json; tail -c 300 $OUT/err3
timeout 300 python3 tools/image_kinds.py --workload C5 --samples 10000 --kernels 6 7 8 --alts 16 --mix 0.7 0.15 0.15 0 0 0 > $OUT/kinds_C5_16alts.json 2> $OUT/err4; cat $OUT/kinds_C5_16alts.json; tail -c 300 $OUT/err4
