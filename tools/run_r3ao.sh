mkdir -p gpurun_out/r3ao
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_kernels.py tests/test_gpu_whole_cohorts.py -x -q 2>&1 | tail -2
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=4,launches=1" "kernel=4,phase=12" "kernel=4,phase=8" "kernel=4,phase=32" 2>&1 | grep kernel= | tee gpurun_out/r3ao/ab_c3.txt
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=4,launches=1" "kernel=4,phase=32" "kernel=4,phase=16" 2>&1 | grep kernel= | tee gpurun_out/r3ao/ab_c2.txt
timeout 1500 python tools/ab.py --workload C3 --samples 10000 --rounds 4 "kernel=4" "kernel=4,launches=1" "kernel=4,phase=12" "kernel=4,phase=8" 2>&1 | grep kernel= | tee gpurun_out/r3ao/ab_c3_whole.txt
