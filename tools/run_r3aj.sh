mkdir -p gpurun_out/r3aj
timeout 800 python -m pytest tests/test_gpu_device_build.py tests/test_gpu_device_build_fasta.py tests/test_gpu_device_build_fuzz.py -q 2>&1 | tail -3
python - <<'PY'
import ctypes, statistics, numpy as np, torch, time
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
ctx = Context(0)
for preset, samples in (("C3", 2000), ("C4", 313)):
    c = Cohort.preset(preset, n_samples=samples)
    ctx.upload_proteome(c.proteome())
    stream = c.txstream(0, c.n_haplotypes, n_threads=64)
    for kernel, window in ((4, 4096), (5, 6144), (5, 7168), (5, 8192), (5, 10240), (2, 32768)):
        b = ctx.batch()
        try:
            ms = b.build_on_device(stream, window, kernel)
        except Exception as e:
            print(preset, kernel, window, "refused:", str(e)[:60]); b.close(); continue
        cn = b.counts()
        t = []
        for r in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); b.execute(); b.sync(); t.append((time.perf_counter() - t0) * 1e3)
        print(preset, "kernel", kernel, "window", window, "build %.2f ms" % ms, "chunks", cn["n_chunks"], "desc", cn["n_desc"], "execute %.3f ms (min %.3f)" % (statistics.median(t[1:]), min(t[1:])))
        b.close()
    host = c.pack(0, c.n_haplotypes, n_threads=64)
    b = ctx.batch(); b.set_packed(host.desc, host.chunks, host.payload, host.hap_out_begin); b.finalize()
    t = []
    for r in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); b.execute(); b.sync(); t.append((time.perf_counter() - t0) * 1e3)
    print(preset, "host-packed", "chunks", host.chunks.shape[0], "execute %.3f ms (min %.3f)" % (statistics.median(t[1:]), min(t[1:])))
    b.close(); stream.close()
PY
