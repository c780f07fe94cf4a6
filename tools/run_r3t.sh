mkdir -p gpurun_out/profiles_r03
cp profiles/traffic_latest.json gpurun_out/profiles_r03/ 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/profiles_r03/pytest_gpu.txt 2>&1; tail -2 gpurun_out/profiles_r03/pytest_gpu.txt
timeout 900 bash tools/profile_round.sh r03_C2 > gpurun_out/profiles_r03/prof_C2.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C3 --workload C3 > gpurun_out/profiles_r03/prof_C3.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C4 --workload C4 > gpurun_out/profiles_r03/prof_C4.txt 2>&1
SKIP_CEILING=1 timeout 900 bash tools/profile_round.sh r03_C5 --workload C5 > gpurun_out/profiles_r03/prof_C5.txt 2>&1
timeout 1500 python bench.py > gpurun_out/profiles_r03/r03_bench_default_line.json 2> gpurun_out/profiles_r03/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/profiles_r03/r03_bench_default_line.json').read().strip().splitlines()[-1])
print('C2', j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'])
d=j.get('device_image_build',{}); print(' device-built', d.get('kernel_choice'), d.get('window_bytes'), d.get('build_kernels_ms'), d.get('execute_ms_device_built_image'), d.get('digests_equal_host_built_image'))
ns=j.get('north_star_cohort',{}); print('C3 whole', ns.get('ms'), ns.get('frac'), ns.get('every_haplotype'))
d=ns.get('device_image_build',{}); print(' device-built', d.get('kernel_choice'), d.get('window_bytes'), d.get('build_kernels_ms'), d.get('execute_ms_device_built_image'), d.get('digests_equal_host_built_image'))
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'])
for w in ('C2','C3','C4','C5'):
    s=json.load(open(f'gpurun_out/profiles_r03/r03_{w}_summary.json'))
    b=s.get('bench_line_under_profiler') or {}
    print(w, 'rocprof step ms', s.get('kernel_ms_per_step'), 'bench', b.get('roofline',{}).get('kernel_ms_avg'), 'frac', b.get('roofline',{}).get('frac'), 'traffic', (s.get('hbm') or {}).get('traffic_bytes_per_step'), 'hbm_min', b.get('roofline',{}).get('hbm_bytes_min_per_launch'))
PY
