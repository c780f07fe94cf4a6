mkdir -p gpurun_out/r3al
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
for w in "C3 2000" "C2 1000" "C4 313"; do set -- $w
for i in 1 2; do
timeout 600 python tools/ab.py --workload $1 --samples $2 --rounds 8 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$1 new  /" | tee -a gpurun_out/r3al/ab.txt
timeout 600 python tools/ab.py --lib build_ab/wave_head.so --workload $1 --samples $2 --rounds 8 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$1 head /" | tee -a gpurun_out/r3al/ab.txt
done; done
