mkdir -p gpurun_out/r3am
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 12 "kernel=4" "kernel=4,rep=0" "kernel=4,nt=1" 2>&1 | grep kernel= | tee gpurun_out/r3am/order_c3.txt
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 12 "kernel=4,linecut=0" "kernel=4" "kernel=4,cut_align=16" "kernel=4,linecut=0,nt=1" 2>&1 | grep kernel= | tee gpurun_out/r3am/ab_c3b.txt
timeout 900 python tools/ab.py --workload C4 --samples 313 --rounds 12 "kernel=4,cut_align=16" "kernel=4,linecut=0" "kernel=4" "kernel=4,cut_align=16,nt=1" 2>&1 | grep kernel= | tee gpurun_out/r3am/ab_c4b.txt
