mkdir -p gpurun_out/r3ah
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4" "kernel=4,wpg=1" "kernel=4,wpg=2" "kernel=4,phase=32" "kernel=4,phase=48" "kernel=4,phase=96" "kernel=4,phase=128" 2>&1 | grep kernel= | tee gpurun_out/r3ah/ab_c3.txt
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=4" "kernel=4,wpg=1" "kernel=4,wpg=2" "kernel=4,phase=48" "kernel=4,phase=96" "kernel=4,phase=128" 2>&1 | grep kernel= | tee gpurun_out/r3ah/ab_c2.txt
