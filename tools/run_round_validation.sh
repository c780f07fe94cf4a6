# Run on the GPU box (gpurun): the round's final validation -- full GPU suite, rocprofv3 stats + PMC traffic per workload (profile_round.sh),
# the default bench line.  Outputs under gpurun_out/profiles_r03/; the summaries are copied to profiles/ by hand.
mkdir -p gpurun_out/profiles_r03
cp profiles/traffic_latest.json gpurun_out/profiles_r03/ 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/profiles_r03/pytest_gpu.txt 2>&1; tail -2 gpurun_out/profiles_r03/pytest_gpu.txt
timeout 900 bash tools/profile_round.sh r03_C2 > gpurun_out/profiles_r03/prof_C2.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C3 --workload C3 > gpurun_out/profiles_r03/prof_C3.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C4 --workload C4 > gpurun_out/profiles_r03/prof_C4.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C5 --workload C5 > gpurun_out/profiles_r03/prof_C5.txt 2>&1
timeout 1500 python bench.py > gpurun_out/profiles_r03/r03_bench_default_line.json 2> gpurun_out/profiles_r03/bench_default.err; echo "bench rc=$?"
