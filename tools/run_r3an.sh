mkdir -p gpurun_out/r3an
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=4,phase=12" "kernel=4,phase=16" "kernel=4,phase=24" "kernel=4,phase=32" 2>&1 | grep kernel= | tee gpurun_out/r3an/ab_c3_slice.txt
timeout 1500 python tools/ab.py --workload C3 --samples 10000 --rounds 4 "kernel=4" "kernel=4,phase=12" "kernel=4,phase=20" 2>&1 | grep kernel= | tee gpurun_out/r3an/ab_c3_whole3.txt
