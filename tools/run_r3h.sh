mkdir -p gpurun_out/r3h
for w in C3:2000 C2:1000 C4:313; do
  wl=${w%%:*}; ns=${w##*:}
  for sub in 0 8192 16384 32768 65536 131072; do
    V2P_WAVE_SUB=$sub timeout 600 python tools/ab.py --workload $wl --samples $ns --rounds 6 "kernel=4" 2>&1 | grep "kernel=4" | sed "s/^/$wl sub=$sub /" >> gpurun_out/r3h/sub.txt
  done
  timeout 600 python tools/ab.py --workload $wl --samples $ns --rounds 6 "kernel=0" 2>&1 | grep "kernel=0" | sed "s/^/$wl old /" >> gpurun_out/r3h/sub.txt
done
cat gpurun_out/r3h/sub.txt
