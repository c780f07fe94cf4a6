# Run on the GPU box (gpurun): the round's validation -- the GPU test suite and smoke(), rocprofv3 stats + PMC traffic of the headline bench
# command (C3 whole) and of C4 whole / C5 whole / C2, the decode profile, the default bench line.  Outputs under gpurun_out/profiles_r06/ (and
# gpurun_out/prof_*); the summaries are copied to profiles/ by hand.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles_r06
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/profiles_r06/pytest_gpu.log 2>&1; tail -3 gpurun_out/profiles_r06/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cp profiles/traffic_latest.json gpurun_out/profiles_r06/ 2>/dev/null
export PROFILES_OUT=profiles_r06
bash tools/profile_round.sh r06_C3whole > gpurun_out/profile_C3whole.log 2>&1; tail -3 gpurun_out/profile_C3whole.log
SKIP_CEILING=1 bash tools/profile_round.sh r06_C4whole --workload C4 --samples 2504 > gpurun_out/profile_C4whole.log 2>&1
SKIP_CEILING=1 bash tools/profile_round.sh r06_C5whole --workload C5 --samples 50000 > gpurun_out/profile_C5.log 2>&1
SKIP_CEILING=1 bash tools/profile_round.sh r06_C2 --workload C2 --samples 1000 > gpurun_out/profile_C2.log 2>&1
bash tools/profile_decode.sh > gpurun_out/profile_decode.log 2>&1
timeout 1500 python bench.py > gpurun_out/profiles_r06/r06_bench_default_line.json 2> gpurun_out/profiles_r06/bench_default.err; echo "bench rc=$?"
timeout 900 python tools/strong_scaling_probe.py > gpurun_out/profiles_r06/r06_strong_scaling_probe.json 2> gpurun_out/profiles_r06/strong_probe.err; grep '^{' gpurun_out/profiles_r06/strong_probe.err
