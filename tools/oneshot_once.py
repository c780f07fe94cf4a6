#!/usr/bin/env python3
"""ONE v2p_batch_build_and_execute on a resident stream of a preset cohort (what rocprofv3 profiles in tools/pmc_parse.sh):
    python3 tools/oneshot_once.py --workload C5 --samples 10000 [--kernel 0] [--reps 1]
Prints one JSON line (v2p_batch_oneshot_info of the last call, the image's counts and form)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--executes", type=int, default=0, help="v2p_batch_execute calls behind the one call")
    a = ap.parse_args()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    c = Cohort.preset(a.workload, n_samples=a.samples)
    nt = max(1, min(64, os.cpu_count() or 1))
    st = c.txstream(0, c.n_haplotypes, n_threads=nt)
    with Context(0) as ctx:
        ctx.upload_proteome(c.proteome())
        rs = ctx.upload_stream(st)
        st.close()
        b = ctx.batch()
        info = None
        for _ in range(a.reps):
            b.reset() if info else None
            b.build_and_execute(rs, a.kernel, 0)
            b.sync()
            info = b.oneshot_info()
        for _ in range(a.executes):
            b.execute()
        b.sync()
        print(json.dumps({"workload": a.workload, "samples": a.samples, "oneshot": {k: v for k, v in info.items() if k != "slice_build_ms"}, "counts": b.counts(), "form": b.image_form()}))
        b.close(); rs.close()


if __name__ == "__main__":
    main()
