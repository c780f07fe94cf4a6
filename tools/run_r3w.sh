mkdir -p gpurun_out/r3w
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=0,phase=0" "kernel=0" "kernel=0,phase=96" "kernel=0,phase=128" "kernel=0,phase=192" > gpurun_out/r3w/ab_c2.txt 2>&1
tail -6 gpurun_out/r3w/ab_c2.txt
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=0,phase=0" "kernel=4,phase=0" "kernel=4" "kernel=4,phase=96" "kernel=4,phase=128" "kernel=1" > gpurun_out/r3w/ab_c3.txt 2>&1
tail -7 gpurun_out/r3w/ab_c3.txt
timeout 900 python tools/ab.py --workload C4 --samples 313 --rounds 8 "kernel=0,phase=0" "kernel=4,phase=0" "kernel=4" "kernel=4,phase=128" "kernel=1" > gpurun_out/r3w/ab_c4.txt 2>&1
tail -6 gpurun_out/r3w/ab_c4.txt
timeout 1200 python -m pytest tests -m gpu -q -x tests/test_gpu_wave_kernel.py tests/test_gpu_kernels.py tests/test_gpu_whole_cohorts.py > gpurun_out/r3w/pytest.txt 2>&1; tail -3 gpurun_out/r3w/pytest.txt
