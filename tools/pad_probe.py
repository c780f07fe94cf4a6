#!/usr/bin/env python3
"""Padded vs dense wave image of the one call (v2p_set_launch_opts variant 0 / 22): same chunk ORDER?  steady execute of each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vcf2prot_amd import build
build.build_hip(); build.build_cohort()
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
wl, samples = sys.argv[1], int(sys.argv[2])
VARS = tuple(int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,22").split(","))
cohort = Cohort.preset(wl, n_samples=samples)
n = cohort.n_haplotypes
stream = cohort.txstream(0, n, n_threads=min(64, os.cpu_count() or 1))
out = {}
with Context(0, development=True) as ctx:      # (the A/B switches live in libv2p_bench.so)
    ctx.upload_proteome(cohort.proteome())
    rs = ctx.upload_stream(stream); stream.close()
    bs = {}
    for var in VARS:
        ctx.set_launch_opts(variant=var)
        b = ctx.batch(); b.build_and_execute(rs, 0, 1); b.sync(); bs[var] = b
    ctx.set_launch_opts()
    if len(VARS) == 2:
        d0, c0, h0 = bs[VARS[0]].download_image(); d1, c1, h1 = bs[VARS[1]].download_image()
        out["desc_equal"] = bool(np.array_equal(d0, d1)); out["chunks_equal_in_order"] = bool(np.array_equal(c0, c1)); out["n_chunks"] = int(c0.shape[0])
        if not out["chunks_equal_in_order"]:
            diff = np.nonzero((c0 != c1).any(axis=1))[0]
            out["first_diff"] = int(diff[0]); out["n_diff"] = int(diff.size)
        out["digests_equal"] = bool(np.array_equal(bs[VARS[0]].digests(), bs[VARS[1]].digests()))
    for rep in range(5):
        for var in VARS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5): bs[var].execute()
            bs[var].sync(); e1.record(); e1.synchronize()
            out.setdefault(f"exec5_ms_v{var}", []).append(round(e0.elapsed_time(e1) / 5, 3))
print(json.dumps(out))
