mkdir -p gpurun_out/profiles_r03
cp profiles/traffic_latest.json gpurun_out/profiles_r03/ 2>/dev/null
timeout 1500 python bench.py > gpurun_out/profiles_r03/r03_bench_default_line.json 2> gpurun_out/profiles_r03/bench_default.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/profiles_r03/bench_default.err
cat gpurun_out/profiles_r03/r03_bench_default_line.json
timeout 900 bash tools/profile_round.sh r03_C2 > gpurun_out/profiles_r03/prof_C2.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C3 --workload C3 > gpurun_out/profiles_r03/prof_C3.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C4 --workload C4 > gpurun_out/profiles_r03/prof_C4.txt 2>&1
tail -n 50 gpurun_out/profiles_r03/prof_C2.txt
