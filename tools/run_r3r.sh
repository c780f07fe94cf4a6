timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_device_build.py -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 600 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=4,chunk_bytes=8192" "kernel=1" 2>&1 | grep "kernel="
done
timeout 600 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=4,chunk_bytes=8192" "kernel=2" 2>&1 | grep "kernel="
