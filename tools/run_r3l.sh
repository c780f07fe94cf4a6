mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3l/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3l/pytest.txt
tail -15 gpurun_out/r3l/pytest.txt
timeout 900 python bench.py --steps 50 > gpurun_out/r3l/bench_c2.json 2> gpurun_out/r3l/bench_c2.err; tail -c 1500 gpurun_out/r3l/bench_c2.json; tail -3 gpurun_out/r3l/bench_c2.err
