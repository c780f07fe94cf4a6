#!/usr/bin/env python3
"""Kernel time of the library-owned batch path (v2p_batch_set_packed/finalize/execute), to check it
matches bench.py's raw-launcher number (same kernel, library-allocated buffers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
c = Cohort.preset("C2", n_samples=int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
img = c.pack(0, c.n_haplotypes, n_threads=64)
with Context(0) as ctx:
    ctx.upload_proteome(c.proteome())
    b = ctx.batch()
    b.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
    b.finalize()
    for _ in range(3):
        b.execute()
    b.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        b.execute()
    b.sync()
    print("library batch path: %.3f ms per pass" % ((time.perf_counter() - t0) * 1e3 / 20))
    b.close()
