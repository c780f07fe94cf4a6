python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --steps 10 "$@" 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$*','ms',round(r['kernel_ms_avg'],3),'min',round(r['kernel_ms_min'],3),'aa/s %.3e'%j['value'],'chunks',j['config']['chunks_per_gpu'])"; }
run --fasta
run --workload C3 --fasta
