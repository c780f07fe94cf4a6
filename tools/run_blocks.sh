timeout 1500 python tools/ab.py --workload C3 --samples 10000 --rounds 4 "kernel=4,phase=24" "kernel=4,phase=28" "kernel=4,phase=32" "kernel=4,phase=36" "kernel=4,phase=40" 2>&1 | grep kernel=
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 6 "kernel=4,phase=20" "kernel=4,phase=28" "kernel=4,phase=32" "kernel=4,phase=40" 2>&1 | grep kernel=
