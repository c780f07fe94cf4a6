#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel-trace stats of the DEVICE image build (tools/build_bench.py) per workload.
# usage: tools/profile_build.sh <tag> [rows] ; outputs gpurun_out/prof_build_<tag>/
set -u
TAG=${1:-r04}
ROWS=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_build_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
for W in "C2 1000 6" "C3 10000 6" "C5 10000 7"; do
  set -- $W
  K=""; [ -n "$ROWS" ] && K="--kernel $3"
  timeout 900 python3 tools/build_bench.py --workload $1 --samples $2 --reps 3 --check $K > $OUT/$1_plain.json 2> $OUT/$1_plain.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1_trace -o t -- python3 tools/build_bench.py --workload $1 --samples $2 --reps 3 $K > $OUT/$1_traced.json 2> $OUT/$1_traced.err
  cp $OUT/$1_trace/*/t_kernel_stats.csv $OUT/$1_kernel_stats.csv 2>/dev/null || cp $OUT/$1_trace/t_kernel_stats.csv $OUT/$1_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/$1_trace
done
tail -n 3 $OUT/*_plain.json
