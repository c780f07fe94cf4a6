mkdir -p gpurun_out/r3q
for args in "C3 11 5" "C1 0 8" "C5 9 30"; do timeout 300 python tools/wave_debug.py --lib build_ab/wave_check.so $args 2>&1 | tail -3; done
timeout 1500 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_batched.py::test_c3_whole_cohort_on_one_gpu_every_haplotype_by_digest --deselect tests/test_gpu_whole_cohorts.py 2>&1 | tail -3
for i in 1 2; do
timeout 600 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=1" 2>&1 | grep "kernel=" | sed "s/^/v3 /"
timeout 600 python tools/ab.py --lib build_ab/wave_pf0.so --workload C2 --samples 1000 --rounds 8 "kernel=4" 2>&1 | grep "kernel=" | sed "s/^/v2 /"
done
timeout 600 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=2" 2>&1 | grep "kernel=" | sed "s/^/v3 /"
timeout 600 python tools/ab.py --lib build_ab/wave_pf0.so --workload C3 --samples 2000 --rounds 8 "kernel=4" 2>&1 | grep "kernel=" | sed "s/^/v2 /"
timeout 600 python tools/ab.py --workload C4 --samples 313 --rounds 8 "kernel=4" "kernel=2" 2>&1 | grep "kernel=" | sed "s/^/v3 /"
