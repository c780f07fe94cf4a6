python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --steps 10 "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*','ms',round(j['roofline']['kernel_ms_avg'],3),'min',round(j['roofline']['kernel_ms_min'],3),'frac',round(j['roofline']['frac'],3), j['verified'] is not None, j['config']['chunks_per_gpu'])"; }
run --xcd-order 1
run --xcd-order 0
run --xcd-order 1 --dbg 1
run --xcd-order 1 --dbg 2
run --xcd-order 1 --max-blocks 2048
