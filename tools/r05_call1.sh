#!/bin/bash
# round 5, GPU call 1: the whole -m gpu suite (new: full-size parity of the device-built images, the 4 GiB haplotype, the new digest),
# then the first-execute scenarios and a GRBM_GUI_ACTIVE pass over build + executes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_call1
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python3 tools/first_execute.py > $OUT/first_execute.json 2> $OUT/first_execute.err
tail -c 600 $OUT/first_execute.err
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_gui -o gui -- python3 tools/build_bench.py --workload C3 --samples 10000 --reps 2 --exec-reps 8 > $OUT/pmc_gui.log 2>&1
tail -c 300 $OUT/pmc_gui.log
ls -la $OUT $OUT/pmc_gui/* | head -30
