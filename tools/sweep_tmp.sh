python tools/ab.py --rounds 12 "dbg=0" "strided=1" "dbg=0" "strided=1" | tail -4
python tools/ab.py --workload C5 --samples 10000 --rounds 8 "dbg=0" "strided=1" | tail -2
