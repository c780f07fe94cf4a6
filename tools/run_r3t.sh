mkdir -p gpurun_out/profiles_r03
cp profiles/traffic_latest.json gpurun_out/profiles_r03/ 2>/dev/null
timeout 900 bash tools/profile_round.sh r03_C2 > gpurun_out/profiles_r03/prof_C2.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C3 --workload C3 > gpurun_out/profiles_r03/prof_C3.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C4 --workload C4 > gpurun_out/profiles_r03/prof_C4.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C5 --workload C5 > gpurun_out/profiles_r03/prof_C5.txt 2>&1
cp gpurun_out/profiles_r03/traffic_latest.json profiles/traffic_latest.json
timeout 1500 python bench.py > gpurun_out/profiles_r03/r03_bench_default_line.json 2> gpurun_out/profiles_r03/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/profiles_r03/r03_bench_default_line.json').read().strip().splitlines()[-1])
print('C2', j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['roofline']['traffic_source'])
ns=j.get('north_star_cohort',{}); print('C3 whole', ns.get('ms'), ns.get('frac'), ns.get('kernel'), ns.get('every_haplotype'))
print(j.get('device_image_build'))
for w in ('C2','C3','C4','C5'):
    s=json.load(open(f'gpurun_out/profiles_r03/r03_{w}_summary.json'))
    print(w, s.get('kernel_ms_per_step'), s.get('launches_per_step'), s.get('hbm'), s.get('l2_hit_rate'), (s.get('bench_line_under_profiler') or {}).get('roofline',{}).get('kernel_ms_avg'))
PY
