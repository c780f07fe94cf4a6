mkdir -p gpurun_out/r3c
for i in 1 2 3; do timeout 300 python tools/wave_debug.py C3 11 5 > gpurun_out/r3c/dbg_prod_C3_$i.txt 2>&1; done
timeout 300 python tools/wave_debug.py C3 11 5 --threads 1 > gpurun_out/r3c/dbg_prod_C3_t1.txt 2>&1
timeout 300 python tools/wave_debug.py C4 2 2 > gpurun_out/r3c/dbg_prod_C4.txt 2>&1
tail -n 12 gpurun_out/r3c/dbg_*.txt
