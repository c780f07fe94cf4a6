mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_kernels.py -x -q > gpurun_out/r3a/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.txt
tail -5 gpurun_out/r3a/pytest.txt
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 10 "kernel=0" "kernel=4" "kernel=4,wpg=1" "kernel=4,wpg=2" > gpurun_out/r3a/ab_c2.txt 2>&1
cat gpurun_out/r3a/ab_c2.txt | tail -6
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 10 "kernel=0" "kernel=1" "kernel=4" "kernel=4,wpg=2" > gpurun_out/r3a/ab_c3.txt 2>&1
cat gpurun_out/r3a/ab_c3.txt | tail -6
timeout 600 python tools/ab.py --workload C4 --samples 313 --rounds 10 "kernel=0" "kernel=4" "kernel=4,wpg=2" > gpurun_out/r3a/ab_c4.txt 2>&1
cat gpurun_out/r3a/ab_c4.txt | tail -5
