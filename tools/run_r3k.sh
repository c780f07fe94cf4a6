mkdir -p gpurun_out/r3k
for w in C2:1000 C3:2000; do
  wl=${w%%:*}; ns=${w##*:}
  for lib in prod wave_aux18 wave_aux16 wave_aux3 wave_aux19 prod; do
    if [ $lib = prod ]; then L=""; else L="--lib build_ab/$lib.so"; fi
    timeout 600 python tools/ab.py $L --workload $wl --samples $ns --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$wl $lib /" >> gpurun_out/r3k/aux.txt
  done
done
cat gpurun_out/r3k/aux.txt
