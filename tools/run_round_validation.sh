# Run on the GPU box (gpurun): the round's validation -- rocprofv3 stats + PMC traffic of the headline bench command (C3 whole), the
# build profile per workload, the default bench line.  Outputs under gpurun_out/profiles_r04/; the summaries are copied to profiles/ by hand.
mkdir -p gpurun_out/profiles_r04
cp profiles/traffic_latest.json gpurun_out/profiles_r04/ 2>/dev/null
timeout 1800 bash tools/profile_round.sh r04_C3whole > gpurun_out/profiles_r04/prof_C3whole.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r04_C2 --workload C2 > gpurun_out/profiles_r04/prof_C2.txt 2>&1
timeout 1500 python bench.py > gpurun_out/profiles_r04/r04_bench_default_line.json 2> gpurun_out/profiles_r04/bench_default.err; echo "bench rc=$?"
