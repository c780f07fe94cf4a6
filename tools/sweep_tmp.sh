python tools/ab.py --workload C5 --samples 10000 --rounds 8 "dbg=6" "dbg=7" | tail -2
