mkdir -p gpurun_out/r3ae
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=4,sync=1" "kernel=4,gap=5" "kernel=4,gap=20" "kernel=4,gap=50" "kernel=4,phase=16,gap=10" "kernel=4,phase=32,gap=10" 2>&1 | grep kernel= | tee gpurun_out/r3ae/ab_c2.txt
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=4,sync=1" "kernel=4,gap=5" "kernel=4,gap=20" "kernel=4,gap=50" 2>&1 | grep kernel= | tee gpurun_out/r3ae/ab_c3.txt
