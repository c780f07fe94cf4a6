python tools/ab.py --workload C3 --samples 2000 --rounds 8 "dbg=0" "dbg=4" | tail -2
python tools/ab.py --workload C5 --samples 10000 --rounds 8 "dbg=0" "dbg=4" | tail -2
python tools/ab.py --rounds 6 "fasta=1" "fasta=1,dbg=3" | tail -2
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
