mkdir -p gpurun_out/r3ak
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3ak/pytest.txt 2>&1; tail -3 gpurun_out/r3ak/pytest.txt
timeout 1500 python bench.py > gpurun_out/r3ak/bench_default.json 2> gpurun_out/r3ak/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r3ak/bench_default.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3ak/bench_default.json').read().strip().splitlines()[-1])
print('C2', j['ms_per_step'], j['roofline']['frac'])
print(j.get('device_image_build'))
ns=j.get('north_star_cohort',{}); print('C3 whole', ns.get('ms'), ns.get('frac'), ns.get('every_haplotype'))
print(ns.get('device_image_build'))
PY
