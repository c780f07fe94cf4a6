for i in 1 2; do
python tools/lib_batch_bench.py
python bench.py --no-cpu-baseline --no-verify --steps 20 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench ms_per_step', round(j['ms_per_step'],3), 'kernel avg', round(j['roofline']['kernel_ms_avg'],3))"
done
