#!/usr/bin/env python3
"""Image kinds side by side on ONE cohort in one process: for each kernel (6 wave rows image, 7 dense rows image, 8 patch image) the build kernels'
time from the resident stream and the steady execute (median of `--rounds`, alternating), every image's haplotype digests equal.

    python tools/image_kinds.py --workload C5 --samples 10000 --kernels 7 8 [--mix 1 0 0 0 0 0] [--alts 64]
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C5")
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--kernels", type=int, nargs="+", default=[7, 8])
    ap.add_argument("--mix", type=float, nargs=6, default=None, help="missense ins del frameshift stop_gained stop_lost")
    ap.add_argument("--alts", type=int, default=0, help="alterations per transcript (alts_fixed)")
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    import torch
    from vcf2prot_amd import build
    build.build_hip(); build.build_cohort()
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    over = {"n_samples": a.samples}
    if a.mix:
        over["mix"] = a.mix
    if a.alts:
        over["alts_fixed"] = a.alts
    c = Cohort.preset(a.workload, **over)
    n = c.n_haplotypes
    st = c.txstream(0, n, n_threads=min(64, os.cpu_count() or 1))
    out = {"workload": a.workload, "samples": a.samples, "mix": a.mix, "alts": a.alts, "tasks": st.n_tasks, "kinds": {}}
    ts = torch.cuda.Stream()
    with Context(0, development=True) as ctx:      # (image kinds the product no longer builds: libv2p_bench.so)
        ctx.upload_proteome(c.proteome())
        rs = ctx.upload_stream(st)
        st.close()
        out["result_bytes"] = rs.counts()["out_bytes"]
        ctx.set_stream(ts.cuda_stream)
        batches, dig0 = {}, None
        for k in a.kernels:
            b = ctx.batch()
            try:
                ms = [b.build_from_stream(rs, k) if i == 0 else (b.reset(), b.build_from_stream(rs, k))[1] for i in range(3)]
            except V2PError as e:
                out["kinds"][str(k)] = {"refused": str(e)[:120]}
                b.close()
                continue
            b.execute(); b.sync()
            dig = b.digests()
            if dig0 is None:
                dig0 = dig
            cn = b.counts()
            rec = {"build_kernels_ms": round(sorted(ms)[1], 4), "digests_equal_first_kind": bool(np.array_equal(dig, dig0)), "descriptors_or_segments": cn["n_desc"], "chunks": cn["n_chunks"]}
            if k == 8:
                _, _, _, ns, npat = (None, None, None, *b.download_patch_image()[3:]) if cn["n_chunks"] < 4000 else (None, None, None, cn["n_desc"], -1)
                rec["segments"], rec["patches"] = ns, npat
            out["kinds"][str(k)] = rec
            batches[k] = b
        times = {k: [] for k in batches}
        for _ in range(a.rounds):
            for k, b in batches.items():
                b.execute()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ts); b.execute(); e1.record(ts); b.sync()
                times[k].append(e0.elapsed_time(e1))
        for k, v in times.items():
            out["kinds"][str(k)]["execute_ms"] = round(statistics.median(v), 4)
        for b in batches.values():
            b.close()
        rs.close()
        ctx.set_stream(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
