#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC traffic passes for bench.py (the headline leg only).
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-r05}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
B="--no-cpu-baseline --no-pcie --no-c2 --no-host-packed --no-live-traffic"
[ -n "${SKIP_CEILING:-}" ] || python3 tools/hbm_ceiling.py 16 > $OUT/hbm_ceiling.json 2>$OUT/hbm_ceiling.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $B --verify sample --steps 10 --warmup 3 "$@" > $OUT/bench_trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py $B --no-verify --steps 3 --warmup 1 "$@" > $OUT/bench_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py $B --no-verify --steps 3 --warmup 1 "$@" > $OUT/bench_write.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -o l2 -- python3 bench.py $B --no-verify --steps 3 --warmup 1 "$@" > $OUT/bench_l2.log 2>&1
python3 tools/summarize_profile.py $OUT $TAG $ROOT/gpurun_out/${PROFILES_OUT:-profiles_r05} "$@"
