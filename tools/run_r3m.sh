mkdir -p gpurun_out/r3m
timeout 900 python -m pytest tests/test_gpu_wave_kernel.py tests/test_gpu_device_build.py -x -q 2>&1 | tail -3
for w in C2:1000 C3:2000; do
  wl=${w%%:*}; ns=${w##*:}
  for lib in prod wave_pf0 wave_pf1024 wave_pf16384 prod wave_pf0; do
    if [ $lib = prod ]; then L=""; else L="--lib build_ab/$lib.so"; fi
    timeout 600 python tools/ab.py $L --workload $wl --samples $ns --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$wl $lib /" >> gpurun_out/r3m/pf.txt
  done
done
cat gpurun_out/r3m/pf.txt
