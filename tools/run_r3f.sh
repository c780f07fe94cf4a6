mkdir -p gpurun_out/r3f
for w in C3:2000 C2:1000; do
  wl=${w%%:*}; ns=${w##*:}
  for lib in prod wave_ab1 wave_ab2 wave_ab3 wave_ab4 wave_ab7; do
    if [ $lib = prod ]; then L=""; else L="--lib build_ab/$lib.so"; fi
    timeout 600 python tools/ab.py $L --workload $wl --samples $ns --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$wl $lib /" >> gpurun_out/r3f/ablate.txt
  done
done
cat gpurun_out/r3f/ablate.txt
