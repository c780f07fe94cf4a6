timeout 1200 python -m pytest tests/test_gpu_device_build.py tests/test_gpu_device_build_fasta.py tests/test_gpu_device_build_fuzz.py tests/test_gpu_vcf_to_fasta.py tests/test_gpu_harness.py -q -x 2>&1 | tail -3
python - <<'PY'
import numpy as np, torch, time, statistics
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
ctx = Context(0)
# a byte-for-byte check on an image large enough for MANY blocks: C3 600 haplotypes (1.1 GB -> 17 blocks of 64 MB)
c = Cohort.preset("C3")
ctx.upload_proteome(c.proteome())
for kernel, window in ((4, 4096), (2, 32768)):
    want = c.pack_grid(0, 600, window, kernel)
    stream = c.txstream(0, 600, n_threads=16)
    b = ctx.batch(); b.build_on_device(stream, window, kernel)
    desc, chunks, hb = b.download_image()
    wc = np.ascontiguousarray(want.chunks)
    ctx._lib.v2p_order_chunks_for_xcds(wc.ctypes.data, wc.shape[0], want.desc.ctypes.data, want.desc.size, c.proteome().size)
    print("kernel", kernel, "chunks", chunks.shape[0], "table equal:", bool(chunks.shape == wc.shape and np.array_equal(chunks, wc)), "desc equal:", bool(np.array_equal(desc, want.desc)))
    b.close(); stream.close()
PY
timeout 1500 python bench.py --no-cpu-baseline --no-pcie > gpurun_out/bench_seg.json 2>gpurun_out/bench_seg.err; echo rc=$?
python - <<'PY'
import json
j=json.loads(open("gpurun_out/bench_seg.json").read().strip().splitlines()[-1])
print("C2", j["ms_per_step"], j["roofline"]["frac"])
d=j.get("device_image_build",{}); print(" device-built", d.get("kernel_choice"), d.get("window_bytes"), d.get("build_kernels_ms"), d.get("execute_ms_device_built_image"), d.get("digests_equal_host_built_image"))
ns=j.get("north_star_cohort",{}); print("C3 whole", ns.get("ms"), ns.get("frac"), ns.get("every_haplotype"))
d=ns.get("device_image_build",{}); print(" device-built", d.get("kernel_choice"), d.get("window_bytes"), d.get("build_kernels_ms"), d.get("execute_ms_device_built_image"), d.get("digests_equal_host_built_image"))
PY
