mkdir -p gpurun_out/r3u
timeout 600 python tools/wave_copy_bench.py --phased-only --rounds 5 --json gpurun_out/r3u/phased.json > gpurun_out/r3u/phased.txt 2>&1
cat gpurun_out/r3u/phased.txt
