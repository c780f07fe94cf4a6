"""v2p_stream_upload of whole cohorts (C3, C2, C5), three uploads each; the third is built and executed (its digests XORed: a checksum).
    python tools/upload_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
out = {}
with Context(0) as ctx:
    for wl, ns in (("C3", 10000), ("C2", 1000), ("C5", 50000)):
        c = Cohort.preset(wl, n_samples=ns)
        st = c.txstream(0, c.n_haplotypes, n_threads=64)
        ctx.upload_proteome(c.proteome())
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); rs = ctx.upload_stream(st); ts.append(time.perf_counter() - t0)
            if rep == 2:
                b = ctx.batch(); b.build_and_execute(rs, 0, 0); b.sync(); d = b.digests(); out[wl + "_dig"] = int(np.bitwise_xor.reduce(d)); b.close()
            rs.close()
        out[wl] = {"upload_s": ts, "bytes": int(st.struct.n_tasks) * 13}
        st.close()
print(json.dumps(out))
