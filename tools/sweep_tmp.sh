run() { python bench.py --no-cpu-baseline --steps 10 "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$*','ms',round(r['kernel_ms_avg'],3),'min',round(r['kernel_ms_min'],3))"; }
run --dbg 1
run --dbg 6
run --dbg 7
