mkdir -p gpurun_out/r3z
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3z/pytest.txt 2>&1; tail -3 gpurun_out/r3z/pytest.txt
timeout 1500 python bench.py > gpurun_out/r3z/bench_default.json 2> gpurun_out/r3z/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3z/bench_default.json').read().strip().splitlines()[-1])
print('C2', j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel'])
ns=j.get('north_star_cohort',{}); print('C3 whole', ns.get('ms'), ns.get('frac'), ns.get('kernel'), ns.get('every_haplotype'))
print(j.get('device_image_build'))
PY
timeout 900 python tools/ab.py --workload C3 --samples 2000 --rounds 6 "kernel=0" "grid=4096,kernel=4" "grid=2048,kernel=4" "grid=32768,kernel=2" > gpurun_out/r3z/ab_c3_grid.txt 2>&1; tail -4 gpurun_out/r3z/ab_c3_grid.txt
