timeout 1500 python tools/ab.py --workload C3 --samples 10000 --rounds 4 "kernel=4" "kernel=4,sub=0" "kernel=4,xcd=0" 2>&1 | grep kernel=
timeout 1500 python tools/ab.py --workload C4 --samples 2504 --rounds 3 "kernel=4" "kernel=4,sub=0" 2>&1 | grep kernel=
