mkdir -p gpurun_out/r3aq
for i in 1 2 3; do
for lib in wave_prod wave_aux18 wave_aux16; do
  timeout 600 python tools/ab.py --lib build_ab/$lib.so --workload C3 --samples 2000 --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/C3 $lib /" | tee -a gpurun_out/r3aq/aux.txt
done; done
for i in 1 2; do
for lib in wave_prod wave_aux18 wave_aux16; do
  timeout 600 python tools/ab.py --lib build_ab/$lib.so --workload C2 --samples 1000 --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/C2 $lib /" | tee -a gpurun_out/r3aq/aux.txt
done; done
