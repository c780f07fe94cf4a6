mkdir -p gpurun_out/r3y
for lib in prod wave_ab1 wave_ab2 wave_ab4; do
  if [ $lib = prod ]; then L=""; else L="--lib build_ab/$lib.so"; fi
  timeout 600 python tools/ab.py $L --workload C3 --samples 2000 --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/C3 $lib /" >> gpurun_out/r3y/ablate.txt
  timeout 600 python tools/ab.py $L --workload C2 --samples 1000 --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/C2 $lib /" >> gpurun_out/r3y/ablate.txt
done
cat gpurun_out/r3y/ablate.txt
for ml in 300 200 120; do
timeout 900 python tools/ab.py --workload C3 --samples 2000 --mean-len $ml --rounds 6 "kernel=0,phase=0" "kernel=2" "kernel=4" "kernel=3" 2>&1 | grep kernel= | sed "s/^/len=$ml /" >> gpurun_out/r3y/cross.txt
done
cat gpurun_out/r3y/cross.txt
bash tools/pmc_sq.sh r3y_c3w --workload C3 --kernel 4 > gpurun_out/r3y/pmc_c3w.txt 2>&1
tail -24 gpurun_out/r3y/pmc_c3w.txt
