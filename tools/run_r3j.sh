mkdir -p gpurun_out/r3j
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 6 "kernel=4" "kernel=4,sub=0" "kernel=4,xcd=0" "kernel=4,gwin=64" "kernel=4,gwin=512" "kernel=4,gwin=4096" "kernel=0" "kernel=0,sub=0" 2>&1 | grep "kernel=" > gpurun_out/r3j/order_c3.txt
timeout 900 python tools/ab.py --workload C2 --samples 500 --rounds 6 "kernel=4" "kernel=4,sub=0" "kernel=4,xcd=0" "kernel=4,gwin=512" 2>&1 | grep "kernel=" > gpurun_out/r3j/order_c2.txt
timeout 900 python tools/ab.py --workload C4 --samples 313 --rounds 6 "kernel=4" "kernel=4,sub=0" "kernel=4,xcd=0" "kernel=4,gwin=512" "kernel=0" 2>&1 | grep "kernel=" > gpurun_out/r3j/order_c4.txt
cat gpurun_out/r3j/order_*.txt
