#!/usr/bin/env python3
"""v2p_batch_download of a whole arena into pageable host memory (C3, 4 000 haplotypes: 7.3 GB) -- the one call's way home when the results
are wanted on the host without the stream-fed pipeline.    python tools/download_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
c = Cohort.preset("C3", n_samples=2000)
st = c.txstream(0, c.n_haplotypes, n_threads=64)
with Context(0) as ctx:
    ctx.upload_proteome(c.proteome())
    rs = ctx.upload_stream(st); st.close()
    b = ctx.batch(); b.build_and_execute(rs, 0, 0); b.sync()
    total = b.counts()["out_bytes"]
    out = np.empty(total, dtype=np.uint8); out[::4096] = 0          # (pages touched: not the first-touch cost of the destination)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx._check(b._lib.v2p_batch_download(b._h, 0, total, out.ctypes.data)); ts.append(time.perf_counter() - t0)
    print(json.dumps({"bytes": int(total), "seconds": ts, "GBps": [total / t / 1e9 for t in ts]}))
    b.close(); rs.close()
