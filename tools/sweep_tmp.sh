run() { python bench.py --no-cpu-baseline --steps 10 "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$*','ms',round(r['kernel_ms_avg'],3),'min',round(r['kernel_ms_min'],3),'chunks',j['config']['chunks_per_gpu'],j['verified'] is not None)"; }
run --cut-align 16
run --cut-align 64
run --cut-align 128
run --cut-align 256
run --cut-align 1024
run --cut-align 4096
run --cut-align 128 --dbg 1
