#!/usr/bin/env python3
"""Steady-state execute of a DENSE rows image as the library launches it (variant 0: descriptors staged by the read-ahead where its rule says so)
against read in place (variant 23), over phase sizes (0 = the library's choice):   python tools/stage_probe.py C3 10000 [0,28,36,40,44,48]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vcf2prot_amd import build
build.build_hip(); build.build_cohort()
from vcf2prot_amd.cohort import Cohort
from vcf2prot_amd.engine import Context
wl, samples = sys.argv[1], int(sys.argv[2])
phases = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "28,36,40,44,48").split(",")]
VARS = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "23,0").split(",")]     # 23: read in place, 0: the library's rule, 25: staged whatever the rule says
cohort = Cohort.preset(wl, n_samples=samples)
n = cohort.n_haplotypes
stream = cohort.txstream(0, n, n_threads=min(64, os.cpu_count() or 1))
out = {"workload": wl, "samples": samples}
with Context(0, development=True) as ctx:      # (the A/B switches live in libv2p_bench.so)
    ctx.upload_proteome(cohort.proteome())
    rs = ctx.upload_stream(stream); stream.close()
    b = ctx.batch(); b.build_and_execute(rs, 0, 0); b.sync()
    dig = b.digests()
    for _ in range(4): b.execute()            # (made dense at the first of these)
    b.sync()
    res = {}
    for rep in range(5):
        for ph in phases:
            for var in VARS:
                ctx.set_launch_opts(variant=var, phase_bytes=ph << 20)
                b.execute(); b.sync()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): b.execute()
                b.sync(); e1.record(); e1.synchronize()
                res.setdefault(f"{ {0: 'library', 23: 'in_place', 25: 'staged'}.get(var, 'v%d' % var) }_ph{ph}", []).append(e0.elapsed_time(e1) / 3)
    ctx.set_launch_opts()
    assert np.array_equal(b.digests(), dig)
    out["ms"] = {k: round(sorted(v)[len(v) // 2], 3) for k, v in res.items()}
print(json.dumps(out))
