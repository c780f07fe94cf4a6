mkdir -p gpurun_out/r3ai
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_whole_cohorts.py -x -q 2>&1 | tail -2
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=4,phase=16" "kernel=4,phase=24" "kernel=4,phase=32" "kernel=4,phase=48" 2>&1 | grep kernel= | tee gpurun_out/r3ai/ab_c3.txt
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=4,phase=24" "kernel=4,phase=32" "kernel=4,phase=48" 2>&1 | grep kernel= | tee gpurun_out/r3ai/ab_c2.txt
timeout 900 python tools/ab.py --workload C4 --samples 313 --rounds 8 "kernel=4" "kernel=4,phase=24" "kernel=4,phase=32" "kernel=4,phase=48" 2>&1 | grep kernel= | tee gpurun_out/r3ai/ab_c4.txt
