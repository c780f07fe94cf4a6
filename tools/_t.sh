cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/final
S=$SECONDS
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -3 gpurun_out/final/pytest_gpu.log; echo "tests $((SECONDS-S))s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$SECONDS
timeout 1500 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$? $((SECONDS-S))s"
python - <<'P'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['one_shot']['total_ms'], d['one_shot']['total_ms_gpu_busy_before'])
P
