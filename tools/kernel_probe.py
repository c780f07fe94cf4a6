#!/usr/bin/env python3
"""A minimal target for rocprofv3 counter passes: one cohort, one image kind, `--builds` builds from the resident stream and `--execs` executes.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ... --kernel-trace -- python3 tools/kernel_probe.py --workload C5 --samples 10000 --kernel 8
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C5")
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--builds", type=int, default=2)
    ap.add_argument("--execs", type=int, default=4)
    a = ap.parse_args()
    from vcf2prot_amd import build
    build.build_hip(); build.build_cohort()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    c = Cohort.preset(a.workload, n_samples=a.samples)
    n = c.n_haplotypes
    st = c.txstream(0, n, n_threads=min(64, os.cpu_count() or 1))
    with Context(0) as ctx:
        ctx.upload_proteome(c.proteome())
        rs = ctx.upload_stream(st)
        st.close()
        b = ctx.batch()
        for _ in range(a.builds):
            b.reset()
            ms = b.build_from_stream(rs, a.kernel)
        for _ in range(a.execs):
            b.execute()
        b.sync()
        print({"kernel": a.kernel, "build_ms": ms, **b.counts()})
        b.close(); rs.close()


if __name__ == "__main__":
    main()
