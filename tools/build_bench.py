#!/usr/bin/env python3
"""Time (and, under rocprofv3, profile) the DEVICE image build of a cohort: transcript stream -> descriptors + chunk table in HBM
(v2p_batch_build_on_device), then one execute on a fresh arena and a few warm ones.

    python tools/build_bench.py --workload C3 --samples 10000 [--reps 3] [--no-exec] [--check]

Prints one JSON line: build_ms per rep, first_execute_ms (arena never written before), warm execute ms, the image's counts, and with
--check whether every haplotype digest equals the host-packed image's.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--exec-reps", type=int, default=5)
    ap.add_argument("--no-exec", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare per-haplotype digests with the host-packed image's")
    ap.add_argument("--kernel", type=int, default=0, help="0 = build_plan's choice")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    import torch
    from vcf2prot_amd import build
    build.build_hip(); build.build_cohort()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.txstream import build_plan
    defaults = {"C2": 1000, "C3": 10000, "C4": 313, "C5": 10000}
    samples = a.samples or defaults[a.workload]
    nt = a.threads or min(64, os.cpu_count() or 1)
    cohort = Cohort.preset(a.workload, n_samples=samples)
    n_haps = cohort.n_haplotypes
    t0 = time.perf_counter()
    stream = cohort.txstream(0, n_haps, n_threads=nt)
    t_stream = time.perf_counter() - t0
    sizes = cohort.result_sizes(0, n_haps, n_threads=nt)
    result_bytes = int(sizes.sum())
    plan = [(a.kernel, a.window)] if a.kernel else build_plan(result_bytes / max(stream.n_tasks, 1))
    out = {"workload": a.workload, "samples": samples, "haplotypes": n_haps, "transcripts": stream.n_tx, "tasks": stream.n_tasks,
           "stream_bytes": stream.nbytes, "result_bytes": result_bytes, "stream_generation_s": t_stream}
    with Context(0) as ctx:
        ctx.upload_proteome(cohort.proteome())
        ts = torch.cuda.Stream()
        ctx.set_stream(ts.cuda_stream)
        builds, firsts, warms = [], [], []
        for rep in range(a.reps):
            b = None
            for kernel, window in plan:
                b = ctx.batch()
                try:
                    ms = b.build_on_device(stream, window, kernel)
                    break
                except V2PError as e:
                    b.close(); b = None
                    if e.code != -9 or (kernel, window) == plan[-1]:
                        raise
            plan = [(kernel, window)]
            builds.append(ms)
            cn = b.counts()
            if not a.no_exec:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ts); b.execute(); e1.record(ts); b.sync()
                firsts.append(e0.elapsed_time(e1))
                if rep == a.reps - 1:
                    for _ in range(a.exec_reps):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(ts); b.execute(); e1.record(ts); b.sync()
                        warms.append(e0.elapsed_time(e1))
                    if a.check:
                        dig = b.digests()
            b.close()
            torch.cuda.empty_cache()
        ctx.set_stream(0)
        out.update({"kernel_choice": kernel, "window_bytes": window, "build_kernels_ms": builds, "first_execute_ms": firsts, "warm_execute_ms": warms,
                    "descriptors": cn["n_desc"], "chunks": cn["n_chunks"]})
        if a.check and not a.no_exec:
            img = cohort.pack(0, n_haps, n_threads=nt)
            hb = ctx.batch()
            hb.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
            hb.finalize(); hb.execute(); hb.sync()
            out["digests_equal_host_built_image"] = bool(np.array_equal(hb.digests(), dig))
            out["host_descriptors"], out["host_chunks"] = int(img.desc.size), int(img.chunks.shape[0])
            # the two images executed alternately in ONE process (boxes and processes differ by more than the images do)
            db = ctx.batch()
            db.build_on_device(stream, window, kernel)
            ctx.set_stream(ts.cuda_stream)
            th, td = [], []
            for _ in range(a.exec_reps + 2):
                for bb, acc in ((hb, th), (db, td)):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(ts); bb.execute(); e1.record(ts); bb.sync()
                    acc.append(e0.elapsed_time(e1))
            ctx.set_stream(0)
            out["ab_execute_ms_host_packed"], out["ab_execute_ms_device_built"] = sorted(th[2:])[len(th[2:]) // 2], sorted(td[2:])[len(td[2:]) // 2]
            hb.close(); db.close()
    stream.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
