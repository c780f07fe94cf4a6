#!/usr/bin/env python3
"""Task vectors -> result bytes ONCE: v2p_batch_build_and_execute (the image built slice by slice on a second HIP stream while the slice
before it is stitched) against the two-call form (v2p_batch_build_from_stream, then v2p_batch_execute), on a RESIDENT stream.

    python tools/oneshot_bench.py --workload C3 --samples 10000 --slices 1,2,4,6,8,12,16 [--reps 3]

Every configuration is measured `--reps` times in two regimes: `cold` -- the host sleeps 0.5 s first (the GPU's clocks have fallen back:
tools/first_execute.py) -- and `warm` -- right behind four executes of an image of the same cohort.  total_ms = HIP events from before the
first build kernel to behind the last stitch kernel.  Every result is checked against the one-piece image's per-haplotype digests.
Prints one JSON line.
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
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--slices", default="1,2,4,6,8,12,16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--oracle", action="store_true", help="also check a sample of haplotypes against the oracle")
    ap.add_argument("--variants", default="", help="comma list of v2p_set_launch_opts variants (0 = the product; 20, 21: builder A/B switches): "
                    "the one call with ONE slice under each, interleaved, instead of the slice sweep")
    ap.add_argument("--phase-mb", default="", help="with --variants: comma list of phase sizes (MB of image per phase; 0 = the library's choice) crossed with the variants")
    a = ap.parse_args()
    import torch
    from vcf2prot_amd import build
    build.build_hip(); build.build_cohort()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    defaults = {"C2": 1000, "C3": 10000, "C4": 2504, "C5": 50000}
    samples = a.samples or defaults[a.workload]
    nt = min(64, os.cpu_count() or 1)
    cohort = Cohort.preset(a.workload, n_samples=samples)
    n = cohort.n_haplotypes
    stream = cohort.txstream(0, n, n_threads=nt)
    out = {"workload": a.workload, "samples": samples, "haplotypes": n, "tasks": stream.n_tasks, "stream_bytes": stream.nbytes}
    ts = torch.cuda.Stream()

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    with Context(0, development=True) as ctx:      # (the A/B switches live in libv2p_bench.so)
        ctx.upload_proteome(cohort.proteome())
        t0 = time.perf_counter()
        rs = ctx.upload_stream(stream)
        out["stream_upload_s"] = time.perf_counter() - t0
        stream.close()
        out["result_bytes"] = rs.counts()["out_bytes"]
        ctx.set_stream(ts.cuda_stream)
        # the reference image: one-piece build, its digests, and a warm batch to run in front of the `warm` measurements
        ref = ctx.batch()
        ref.build_from_stream(rs, a.kernel)
        ref.execute(); ref.sync()
        dig = ref.digests()
        cn = ref.counts()
        out["descriptors"], out["chunks"] = cn["n_desc"], cn["n_chunks"]
        if a.oracle:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
            from sir_oracle import COracle
            orc = COracle()
            for h in np.linspace(0, n - 1, 64).astype(int).tolist():
                hap = cohort.haplotype(h)
                t = orc.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
                want = orc.gir_execute_u8(t, cohort.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
                assert int(dig[h]) == orc.digest_u8(want), h
            out["oracle_checked"] = 64

        def regime(warm):
            if warm:
                for _ in range(4):
                    ref.execute()
                ref.sync()
            else:
                time.sleep(0.5)

        b = ctx.batch()
        res = {}
        if a.variants:
            vs = [int(x) for x in a.variants.split(",")]
            phs = [int(x) for x in a.phase_mb.split(",")] if a.phase_mb else [0]
            for rep in range(a.reps):
                for warm in (False, True):
                  for ph in phs:
                    for var in vs:
                        ctx.set_launch_opts(variant=var, phase_bytes=ph << 20)
                        b.reset()
                        regime(warm)
                        b.build_and_execute(rs, a.kernel, 0)        # (0: the library's own choice of slices)
                        b.sync()
                        info = b.oneshot_info()
                        assert np.array_equal(b.digests(), dig), var
                        res.setdefault(f"variant{var}{'_ph%d' % ph if ph else ''}_{'warm' if warm else 'cold'}", []).append(
                            {"total_ms": info["total_ms"], "build_ms_sum": info["build_ms"], "tables_ms": info.get("tables_ms", 0.0), "call_wall_ms": info["call_wall_ms"]})
            # the image each variant leaves behind, re-executed (steady state), interleaved
            steady = {}
            bs = {}
            for var in sorted(set(vs)):
                ctx.set_launch_opts(variant=var)
                bs[var] = ctx.batch()
                bs[var].build_and_execute(rs, a.kernel, 0); bs[var].sync()
            ctx.set_launch_opts()
            for _ in range(3):
                for var in bs:
                    bs[var].execute()
            for var in bs:
                bs[var].sync()
            for _ in range(7):
                for var in bs:
                    e0, e1 = ev()
                    e0.record(ts); bs[var].execute(); e1.record(ts); bs[var].sync()
                    steady.setdefault(f"variant{var}", []).append(e0.elapsed_time(e1))
            for var in bs:
                assert np.array_equal(bs[var].digests(), dig), var
                bs[var].close()
            out["steady_execute_ms"] = {k: round(sorted(v)[len(v) // 2], 3) for k, v in steady.items()}
            out["runs"] = res
            out["summary"] = {k: [round(sorted(r[f] for r in v)[len(v) // 2], 3) for f in ("total_ms", "build_ms_sum", "tables_ms")] for k, v in res.items()}
            b.close(); ref.close(); rs.close()
            ctx.set_stream(0)
            print(json.dumps(out))
            return
        # the two-call form: build, then execute, one after the other
        for warm in (False, True):
            rows = []
            for _ in range(a.reps):
                b.reset()
                regime(warm)
                e0, e1, = ev()
                e0.record(ts)
                ms_b = b.build_from_stream(rs, a.kernel)
                em0, em1 = ev()
                em0.record(ts); b.execute(); em1.record(ts); b.sync()
                e1.record(ts); e1.synchronize()
                assert np.array_equal(b.digests(), dig)
                rows.append({"build_kernels_ms": ms_b, "execute_ms": em0.elapsed_time(em1), "span_ms_incl_host_round_trips": e0.elapsed_time(e1)})
            res[f"two_calls_{'warm' if warm else 'cold'}"] = rows
        for S in [int(x) for x in a.slices.split(",")]:
            for warm in (False, True):
                rows = []
                for _ in range(a.reps):
                    b.reset()
                    regime(warm)
                    b.build_and_execute(rs, a.kernel, S)
                    b.sync()
                    info = b.oneshot_info()
                    assert np.array_equal(b.digests(), dig), S
                    rows.append({"total_ms": info["total_ms"], "build_ms_sum": info["build_ms"], "call_wall_ms": info["call_wall_ms"], "n_slices": info["n_slices"],
                                 "slice_build_ms": [round(x, 3) for x in info["slice_build_ms"]]})
                res[f"one_call_S{S}_{'warm' if warm else 'cold'}"] = rows
        # the image a sliced call leaves behind, re-executed (steady state) next to the one-piece image
        b.reset(); b.build_and_execute(rs, a.kernel, 8); b.sync()
        tb, tr = [], []
        for _ in range(3):
            b.execute(); ref.execute()
        b.sync()
        for _ in range(7):
            for bb, acc in ((b, tb), (ref, tr)):
                e0, e1 = ev()
                e0.record(ts); bb.execute(); e1.record(ts); bb.sync()
                acc.append(e0.elapsed_time(e1))
        out["steady_execute_ms_sliced_image"], out["steady_execute_ms_one_piece_image"] = sorted(tb)[3], sorted(tr)[3]
        out["runs"] = res
        out["summary"] = {k: round(sorted(r.get("total_ms", r.get("build_kernels_ms", 0) + r.get("execute_ms", 0)) for r in v)[len(v) // 2], 3) for k, v in res.items()}
        b.close(); ref.close(); rs.close()
        ctx.set_stream(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
