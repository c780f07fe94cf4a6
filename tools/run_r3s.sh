timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_device_build.py tests/test_gpu_device_build_fuzz.py -x -q 2>&1 | tail -3
timeout 600 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=2" 2>&1 | grep "kernel="
timeout 600 python tools/ab.py --workload C4 --samples 313 --rounds 8 "kernel=4" "kernel=2" 2>&1 | grep "kernel="
timeout 600 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=1" 2>&1 | grep "kernel="
