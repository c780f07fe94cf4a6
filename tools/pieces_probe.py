#!/usr/bin/env python3
"""A dense rows image executed again: from its piece image (the library, dense_pieces.h) against the dense kernel (v2p_set_launch_opts variant 28),
alternating in one process:   python tools/pieces_probe.py C5 10000"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vcf2prot_amd import build
build.build_hip(); build.build_cohort()
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
wl, samples = sys.argv[1], int(sys.argv[2])
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cohort = Cohort.preset(wl, n_samples=samples)
n = cohort.n_haplotypes
stream = cohort.txstream(0, n, n_threads=min(64, os.cpu_count() or 1))
out = {"workload": wl, "samples": samples, "haplotypes": n}
with Context(0, development=True) as ctx:      # (the A/B switches live in libv2p_bench.so)
    ctx.upload_proteome(cohort.proteome())
    rs = ctx.upload_stream(stream); stream.close()
    b = ctx.batch(); b.build_and_execute(rs, kernel, 0); b.sync()
    dig = b.digests()
    cn = b.counts(); out.update({"descriptors": cn["n_desc"], "chunks": cn["n_chunks"], "result_bytes": cn["out_bytes"]})
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.execute(); b.sync(); e1.record(); e1.synchronize()
    out["first_re_execute_ms_incl_conversion"] = round(e0.elapsed_time(e1), 3)
    out["image_form"] = b.image_form()
    res = {}
    for rep in range(7):
        for var in (0, 28):
            ctx.set_launch_opts(variant=var)
            b.execute(); b.sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): b.execute()
            b.sync(); e1.record(); e1.synchronize()
            res.setdefault("pieces" if var == 0 else "dense_kernel", []).append(e0.elapsed_time(e1) / 5)
            assert np.array_equal(b.digests(), dig)
    ctx.set_launch_opts()
    out["ms"] = {k: round(sorted(v)[len(v) // 2], 4) for k, v in res.items()}
print(json.dumps(out))
