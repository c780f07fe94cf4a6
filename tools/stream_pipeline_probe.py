"""Sweep of the stream-fed pipeline (v2p_pipeline_submit_stream) over slice size and slots: bench.py's stream_pipeline_leg on one cohort.
The reference digests are the one call's on the whole cohort (which the bench and the GPU suite compare with the oracle's).

    python tools/stream_pipeline_probe.py [--workload C3] [--samples N] [--slice-mb 2304,1152,576] [--slots 3,4]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--slice-mb", default="2304,1152,576")
    ap.add_argument("--slots", default="3,4")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import bench
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    samples = args.samples or bench.DEFAULT_SAMPLES["strong"][args.workload]
    nt = max(1, min(64, os.cpu_count() or 1))
    c = Cohort.preset(args.workload, n_samples=samples)
    with Context(0) as ctx:
        ctx.upload_proteome(c.proteome())
        st = c.txstream(0, c.n_haplotypes, n_threads=nt)
        rs = ctx.upload_stream(st)
        st.close()
        b = ctx.batch()
        b.build_and_execute(rs, 0, 0); b.sync()
        ref = np.array(b.digests(), dtype=np.uint64)
        b.close(); rs.close()
    rows = []
    for mb in [int(x) for x in args.slice_mb.split(",")]:
        for slots in [int(x) for x in args.slots.split(",")]:
            r = bench.stream_pipeline_leg(args.workload, samples, nt, ref, target_slice_bytes=mb << 20, slots=slots, reps=2, device=0)
            row = {"slice_mb": mb, "slots": slots, "slices": r["slices"], "seconds": r["seconds"], "d2h_GBps": r["d2h_GBps"], "aa_per_s": r["aa_per_s"],
                   "stage_ms": r["stage_ms_per_slice"], "runner_ms": r["runner_ms_per_slice"]}
            print(json.dumps(row), flush=True)
            rows.append(row)
    if args.out:
        json.dump({"workload": args.workload, "samples": samples, "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
