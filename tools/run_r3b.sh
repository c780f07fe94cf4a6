mkdir -p gpurun_out/r3b
for args in "C3 11 5" "C1 0 8" "C4 2 2" "C5 9 30"; do
  timeout 300 python tools/wave_debug.py --lib build_ab/wave_check.so $args > gpurun_out/r3b/dbg_check_$(echo $args | tr ' ' '_').txt 2>&1
  timeout 300 python tools/wave_debug.py $args > gpurun_out/r3b/dbg_prod_$(echo $args | tr ' ' '_').txt 2>&1
done
tail -n 6 gpurun_out/r3b/dbg_*.txt
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py -q > gpurun_out/r3b/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.txt
tail -5 gpurun_out/r3b/pytest.txt
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 10 "kernel=0" "kernel=4" "kernel=4,wpg=1" "kernel=4,wpg=2" > gpurun_out/r3b/ab_c2.txt 2>&1
tail -5 gpurun_out/r3b/ab_c2.txt
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 10 "kernel=0" "kernel=4" "kernel=4,wpg=2" > gpurun_out/r3b/ab_c3.txt 2>&1
tail -4 gpurun_out/r3b/ab_c3.txt
