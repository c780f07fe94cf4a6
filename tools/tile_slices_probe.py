#!/usr/bin/env python3
"""A/B (development library): a TILE image built and executed in S slices -- slice j's tiles executed on a second stream while slice
j + 1 is parsed -- against the product's form (one parse, one execute).  The one call's HIP-event time, median of --reps calls per form,
the forms alternating; every form's digests against the first.
    python3 tools/tile_slices_probe.py --workload C5 --samples 10000 [--slices 0 2 4 8 16]"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C5")
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--slices", type=int, nargs="*", default=[0, 2, 4, 8, 16])
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    from vcf2prot_amd import build
    build.build_all(); build.build_bench()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    c = Cohort.preset(a.workload, n_samples=a.samples)
    nt = max(1, min(64, os.cpu_count() or 1))
    st = c.txstream(0, c.n_haplotypes, n_threads=nt)
    out = {"workload": a.workload, "samples": a.samples, "haplotypes": c.n_haplotypes, "ms": {}, "digests_equal": True}
    with Context(0, development=True) as ctx:
        ctx.upload_proteome(c.proteome())
        rs = ctx.upload_stream(st)
        st.close()
        b = ctx.batch()
        ref = None
        times = {s: [] for s in a.slices}
        for rep in range(a.reps + 1):
            for s in a.slices:
                b.reset()
                b.build_and_execute(rs, 9, s)
                b.sync()
                info = b.oneshot_info()
                if rep:                                   # (the first round allocates)
                    times[s].append(info["total_ms"])
                if rep in (0, a.reps):
                    d = b.digests()
                    if ref is None:
                        ref = d
                    elif not np.array_equal(d, ref):
                        out["digests_equal"] = False
        for s in a.slices:
            out["ms"][str(s)] = {"median": statistics.median(times[s]), "min": min(times[s]), "max": max(times[s])}
        out["counts"] = b.counts(); out["form"] = b.image_form()
        b.close(); rs.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
