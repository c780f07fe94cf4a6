#!/usr/bin/env python3
"""Phase size and store policy of the launcher on a DEVICE-BUILT image (rows image), alternating in one process:

    python tools/phase_sweep.py --workload C3 --samples 10000 [--phases 20 24 28 32 40 64] [--rounds 7]

Prints one JSON line: median ms per (phase MB, sc1) and the library's default."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--phases", type=int, nargs="*", default=[20, 24, 28, 32, 40, 64])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--all-forms", action="store_true", help="with --launch-forms: also one launch for all phases, own read-ahead kernels, no read-ahead")
    ap.add_argument("--launch-forms", action="store_true", help="A/B of the phased launcher's forms (v2p_set_launch_opts variant 16 / 17 / 18) instead of the store policy")
    a = ap.parse_args()
    import torch
    from vcf2prot_amd import build
    build.build_hip(); build.build_cohort()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    from vcf2prot_amd.txstream import build_on_device_auto
    c = Cohort.preset(a.workload, n_samples=a.samples)
    n = c.n_haplotypes
    nt = min(64, os.cpu_count() or 1)
    st = c.txstream(0, n, n_threads=nt)
    rb = int(c.result_sizes(0, n, n_threads=nt).sum())
    with Context(0, development=True) as ctx:      # (the A/B switches live in libv2p_bench.so)
        ctx.upload_proteome(c.proteome())
        b = ctx.batch()
        info = build_on_device_auto(b, st, rb)
        st.close()
        ts = torch.cuda.Stream()
        ctx.set_stream(ts.cuda_stream)
        variants = [("default", {})] + [(f"phase={p},sc1={s}", dict(phase_bytes=p << 20, store_sc1=s)) for p in a.phases for s in (0, 1)]
        if a.launch_forms:      # the launcher's forms on this (rows) image: one launch for all phases, own read-ahead kernels, no read-ahead
            variants = [("default", {})] + [(f"{name},phase={p}", dict(phase_bytes=p << 20, variant=v)) for name, v in (("dual", 19), ("ride", 0)) + ((("one_launch", 16), ("own_touch", 17), ("no_touch", 18)) if a.all_forms else ())
                                            for p in a.phases]
        for _ in range(6):
            b.execute()
        b.sync()
        times = {k: [] for k, _ in variants}
        for _ in range(a.rounds):
            for name, opts in variants:
                ctx.set_launch_opts(**opts)
                b.execute()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ts); b.execute(); e1.record(ts); b.sync()
                times[name].append(e0.elapsed_time(e1))
        ctx.set_launch_opts()
        ctx.set_stream(0)
        b.close()
    print(json.dumps({"workload": a.workload, "samples": a.samples, "kernel": info["kernel"], "result_bytes": rb,
                      "median_ms": {k: round(statistics.median(v), 4) for k, v in times.items()}, "min_ms": {k: round(min(v), 4) for k, v in times.items()}}))


if __name__ == "__main__":
    main()
