mkdir -p gpurun_out/r3n
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r3n/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3n/pytest.txt
tail -25 gpurun_out/r3n/pytest.txt
timeout 1200 python bench.py --steps 50 > gpurun_out/r3n/bench_c2.json 2> gpurun_out/r3n/bench_c2.err; python - <<'PY'
import json
l=json.loads(open('gpurun_out/r3n/bench_c2.json').read().strip().split('\n')[-1])
print({k:l[k] for k in ('value','ms_per_step')}, l['roofline']['frac'], l['roofline']['kernel'], l['roofline']['kernel_ms_avg'])
print('device build', {k:l['device_image_build'].get(k) for k in ('build_kernels_ms','execute_ms_device_built_image','digests_equal_host_built_image','window_bytes')})
print('north star', l.get('north_star_cohort'))
PY
tail -3 gpurun_out/r3n/bench_c2.err
