mkdir -p gpurun_out/r3x
timeout 600 python tools/wave_copy_bench.py --phased-only --rounds 3 --json gpurun_out/r3x/phased.json 2>&1 | grep -v "nt loads" | head -12
timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 8 "kernel=0,phase=0" "kernel=0" "kernel=0,lines=1" "kernel=0,phase=48,lines=1" "kernel=0,phase=32" "kernel=0,phase=16" > gpurun_out/r3x/ab_c2.txt 2>&1
tail -6 gpurun_out/r3x/ab_c2.txt
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 8 "kernel=4,phase=0" "kernel=4" "kernel=4,lines=1" "kernel=4,phase=32" "kernel=4,phase=48"  > gpurun_out/r3x/ab_c3.txt 2>&1
tail -5 gpurun_out/r3x/ab_c3.txt
