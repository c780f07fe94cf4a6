mkdir -p gpurun_out/r3v
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=0,phase=0" "kernel=0" "kernel=0,phase=24" "kernel=0,phase=96" "kernel=1,phase=0" "kernel=1" > gpurun_out/r3v/ab_c2.txt 2>&1
tail -7 gpurun_out/r3v/ab_c2.txt
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=0,phase=0" "kernel=0" "kernel=0,phase=24" "kernel=0,phase=96" "kernel=4,phase=0" "kernel=4" "kernel=4,phase=24" > gpurun_out/r3v/ab_c3.txt 2>&1
tail -8 gpurun_out/r3v/ab_c3.txt
timeout 900 python tools/ab.py --workload C4 --samples 313 --rounds 8 "kernel=0,phase=0" "kernel=0" "kernel=4,phase=0" "kernel=4" > gpurun_out/r3v/ab_c4.txt 2>&1
tail -5 gpurun_out/r3v/ab_c4.txt
timeout 900 python tools/ab.py --workload C5 --samples 6250 --rounds 8 "kernel=0,phase=0" "kernel=0" "kernel=0,phase=24" "kernel=0,phase=96" > gpurun_out/r3v/ab_c5.txt 2>&1
tail -5 gpurun_out/r3v/ab_c5.txt
