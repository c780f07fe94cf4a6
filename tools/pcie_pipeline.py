#!/usr/bin/env python3
"""PCIe-inclusive throughput of the batched engine (not bench.py's `value`): a cohort is cut into
images that are streamed through v2p_pipeline_* -- H2D of 8-byte descriptors + alt bytes, stitch
kernel, D2H of the results into pinned host memory -- with n_slots images in flight.

    python tools/pcie_pipeline.py [--workload C2] [--samples 500] [--images 10] [--slots 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from vcf2prot_amd.cohort import Cohort  # noqa: E402
from vcf2prot_amd.engine import Context, Pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--samples", type=int, default=500)
    ap.add_argument("--images", type=int, default=10)
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--fasta", action="store_true")
    a = ap.parse_args()
    c = Cohort.preset(a.workload, n_samples=a.samples)
    n = c.n_haplotypes
    per = (n + a.images - 1) // a.images
    imgs = [c.pack(h, min(n, h + per), n_threads=min(64, os.cpu_count() or 1), fasta=a.fasta) for h in range(0, n, per)]
    aa = sum(i.n_copy_bytes for i in imgs)
    out_total = sum(i.out_bytes for i in imgs)
    h2d = sum(i.desc.nbytes + i.chunks.nbytes + i.payload.nbytes for i in imgs)
    with Context(0) as ctx:
        if a.fasta:
            ctx.upload_reference(c.proteome(), c.fasta_headers())
        else:
            ctx.upload_proteome(c.proteome())
        res = {}
        for slots in sorted({1, a.slots}):
            pipe = Pipeline(ctx, slots)
            for rep in range(2):                       # first pass allocates and pins
                t0 = time.perf_counter()
                inflight, checksum = [], 0
                for img in imgs:
                    if len(inflight) == slots:
                        t = inflight.pop(0)
                        r = pipe.wait(t)
                        checksum ^= int(r[::4097].sum())
                        pipe.release(t)
                    inflight.append(pipe.submit(img.desc, img.chunks, img.payload, img.out_bytes))
                for t in inflight:
                    r = pipe.wait(t)
                    checksum ^= int(r[::4097].sum())
                    pipe.release(t)
                secs = time.perf_counter() - t0
            res[f"slots_{slots}"] = {"seconds": secs, "aa_per_s": aa / secs, "d2h_GBs": out_total / secs / 1e9}
            pipe.close()
    print(json.dumps({"workload": a.workload, "samples": a.samples, "images": len(imgs), "aa": aa, "h2d_bytes": h2d,
                      "d2h_bytes": out_total, "fasta": a.fasta, **res}))


if __name__ == "__main__":
    main()
