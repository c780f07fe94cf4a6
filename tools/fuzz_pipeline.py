#!/usr/bin/env python3
"""Random irregular transcript streams (tests/stream_util.py), cut into random slices of haplotypes, through the stream-fed pipeline
(v2p_pipeline_submit_stream) with a random number of slots, kernel choices 0 / 7 / 9 (an image kind asked for by number may refuse a
slice: V2P_ERR_UNSUPPORTED from wait, the slice then goes again under the rule), with and without FASTA emit and digests; every
haplotype's HOST bytes against the numpy expectation, every digest against the oracle's function over those bytes.
    python tools/fuzz_pipeline.py [first_seed] [n_seeds]        (FUZZ_TRACE=1: every slice on stderr before it is submitted)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
from vcf2prot_amd import build  # noqa: E402
build.build_hip(); build.build_cohort()
from vcf2prot_amd.engine import Context, Pipeline  # noqa: E402
from vcf2prot_amd._native import V2PError  # noqa: E402
from stream_util import Stream, random_stream  # noqa: E402
from sir_oracle import COracle  # noqa: E402

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 40
orc = COracle()
bad, slices_run, refused = [], 0, 0


def cut(stream, a, b, fasta):
    k = stream.keep
    hb, tb, ab = k[0].astype(np.int64), k[4].astype(np.int64), k[5].astype(np.int64)
    t0, t1 = int(hb[a]), int(hb[b])
    k0, k1, a0, a1 = int(tb[t0]), int(tb[t1]), int(ab[t0]), int(ab[t1])
    return Stream(hb[a:b + 1] - t0, k[1][t0:t1], k[2][t0:t1], k[3][t0:t1], tb[t0:t1 + 1] - k0, ab[t0:t1 + 1] - a0,
                  k[6][k0:k1], k[7][k0:k1], k[8][k0:k1], k[9][k0:k1], k[10][a0:a1], k[11][t0:t1] if fasta else None, k[12][t0:t1] if fasta else None)


with Context(0) as ctx:
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        shape = ("snv", "mix", "long")[seed % 3]
        fasta = seed % 2 == 1
        n_haps = int(rng.integers(1, 500))
        if fasta:
            proteome, headers, stream, want = random_stream(rng, n_haps=n_haps, n_ref_tx=int(rng.integers(1, 40)), shape=shape, window=4096, fasta=True)
            ctx.upload_reference(proteome, headers)
        else:
            proteome, stream, want = random_stream(rng, n_haps=n_haps, n_ref_tx=int(rng.integers(1, 40)), shape=shape, window=4096)
            ctx.upload_proteome(proteome)
        n_cuts = int(rng.integers(0, min(9, n_haps)))
        cuts = [0] + sorted(set(int(x) for x in rng.integers(0, n_haps + 1, size=n_cuts))) + [n_haps]       # (empty slices too: two equal cuts)
        slots = int(rng.integers(1, 5))
        pipe = Pipeline(ctx, slots)
        try:
            inflight = []

            def finish(job):
                global slices_run, refused
                t, a, b, kernel, dig = job
                try:
                    out = pipe.wait(t)
                except V2PError as e:
                    pipe.release(t)
                    if not (kernel in (7, 9) and e.code == -9):
                        raise
                    refused += 1
                    t = pipe.submit_stream(cut(stream, a, b, fasta), 0, dig)
                    assert t >= 0
                    out = pipe.wait(t)
                info = pipe.result_info(t)
                hob = info["hap_out_begin"]
                for h in range(a, b):
                    got = out[int(hob[h - a]):int(hob[h - a + 1])]
                    if got.size != want[h].size or not np.array_equal(got, want[h]):
                        bad.append({"seed": seed, "slice": [a, b], "kernel": kernel, "hap": h}); break
                    if dig and int(info["digests"][h - a]) != orc.digest_u8(np.ascontiguousarray(got)):
                        bad.append({"seed": seed, "slice": [a, b], "kernel": kernel, "hap": h, "what": "digest"}); break
                pipe.release(t)
                slices_run += 1
            for a, b in zip(cuts[:-1], cuts[1:]):
                kernel = int(rng.choice([0, 0, 7, 9]))
                dig = bool(rng.integers(0, 2))
                if os.environ.get("FUZZ_TRACE"):
                    print("cfg", seed, shape, fasta, n_haps, "slice", a, b, "kernel", kernel, "slots", slots, file=sys.stderr, flush=True)
                if len(inflight) == slots:
                    finish(inflight.pop(0))
                t = pipe.submit_stream(cut(stream, a, b, fasta), kernel, dig)
                assert t >= 0
                inflight.append((t, a, b, kernel, dig))
            while inflight:
                finish(inflight.pop(0))
        except Exception as e:                                   # noqa: BLE001
            bad.append({"seed": seed, "error": repr(e)[:300]})
        finally:
            pipe.close()
        if fasta:
            ctx.upload_proteome(proteome)
print(json.dumps({"seeds": [first, first + count], "slices": slices_run, "refused_by_number": refused, "failures": bad[:20], "n_failures": len(bad)}))
sys.exit(1 if bad else 0)
