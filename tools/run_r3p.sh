mkdir -p gpurun_out/r3p
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 600 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=1" 2>&1 | grep "kernel=" | sed "s/^/new /"
timeout 600 python tools/ab.py --lib build_ab/wave_pf0.so --workload C2 --samples 1000 --rounds 8 "kernel=4" 2>&1 | grep "kernel=" | sed "s/^/old /"
done
timeout 600 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=2" 2>&1 | grep "kernel=" | sed "s/^/new /"
