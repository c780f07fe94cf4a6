( time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | grep real
tail -c 600 gpurun_out/bench_default.json
for w in C3 C5; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['config']['workload'][:40], d['ms_per_step'], d['value'], d['roofline']['frac'])"; done
python bench.py --fasta --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fasta', d['ms_per_step'], d['value'], d['roofline']['frac'])"
