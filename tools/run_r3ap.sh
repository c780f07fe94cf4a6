mkdir -p gpurun_out/r3ap
timeout 1500 python tools/ab.py --workload C3 --samples 10000 --rounds 4 "kernel=4" "kernel=4,phase=16" "kernel=4,phase=24" "kernel=4,xcd=0" "kernel=4,wpg=2" 2>&1 | grep kernel= | tee gpurun_out/r3ap/ab_c3_whole.txt
