timeout 900 python tools/ab.py --workload C2 --samples 1000 --rounds 10 "kernel=4" "kernel=4,sc1=1" "kernel=4,nt=1" "kernel=4,sc1=1,nt=1" 2>&1 | grep kernel=
timeout 900 python tools/ab.py --workload C4 --samples 313 --rounds 10 "kernel=4" "kernel=4,sc1=1" 2>&1 | grep kernel=
timeout 900 python tools/ab.py --workload C2 --samples 200 --rounds 10 "kernel=4" "kernel=4,sc1=1" 2>&1 | grep kernel=
