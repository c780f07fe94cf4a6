timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 1500 python bench.py --no-cpu-baseline --no-pcie > gpurun_out/bench_blocks.json 2>gpurun_out/bench_blocks.err; echo rc=$?
python - <<'PY'
import json
j=json.loads(open("gpurun_out/bench_blocks.json").read().strip().splitlines()[-1])
print("C2", j["ms_per_step"], j["roofline"]["frac"])
d=j.get("device_image_build",{}); print(" device-built", d.get("kernel_choice"), d.get("window_bytes"), d.get("build_kernels_ms"), d.get("execute_ms_device_built_image"), d.get("digests_equal_host_built_image"))
ns=j.get("north_star_cohort",{}); print("C3 whole", ns.get("ms"), ns.get("frac"), ns.get("every_haplotype"))
d=ns.get("device_image_build",{}); print(" device-built", d.get("kernel_choice"), d.get("window_bytes"), d.get("build_kernels_ms"), d.get("execute_ms_device_built_image"), d.get("digests_equal_host_built_image"))
PY
