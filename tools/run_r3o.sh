mkdir -p gpurun_out/r3o
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=1" "kernel=4" > gpurun_out/r3o/ab_c2.txt 2>&1; grep kernel= gpurun_out/r3o/ab_c2.txt
for lib in wave_pf0; do timeout 600 python tools/ab.py --lib build_ab/$lib.so --workload C2 --samples 1000 --rounds 8 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$lib /"; done
timeout 2400 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_batched.py::test_c3_whole_cohort_on_one_gpu_every_haplotype_by_digest --durations=5 > gpurun_out/r3o/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3o/pytest.txt
tail -12 gpurun_out/r3o/pytest.txt
