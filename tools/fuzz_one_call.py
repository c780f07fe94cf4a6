#!/usr/bin/env python3
"""A wider net than the fixed seeds of tests/test_gpu_oneshot.py: random irregular transcript streams (tests/stream_util.py) through the ONE
call under every kernel choice and slicing, the image's re-execution forms (dense from padded, pieces from dense, staged) included --
every haplotype's bytes against the numpy expectation.    python tools/fuzz_one_call.py [first_seed] [n_seeds]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vcf2prot_amd import build
build.build_hip(); build.build_cohort()
from vcf2prot_amd.engine import Context
from vcf2prot_amd._native import V2PError
from stream_util import random_stream

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad, runs = [], 0
with Context(0) as ctx:
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        shape = ("snv", "mix", "long")[seed % 3]
        n_haps = int(rng.integers(1, 700)); n_ref = int(rng.integers(1, 40)); window = int(rng.choice([1024, 4096, 8192]))
        fasta = os.environ.get("FUZZ_FASTA") == "1" or (os.environ.get("FUZZ_FASTA") is None and seed % 2 == 1)
        if fasta:
            proteome, headers, stream, want = random_stream(rng, n_haps=n_haps, n_ref_tx=n_ref, shape=shape, window=window, fasta=True)
            ctx.upload_reference(proteome, headers)
        else:
            proteome, stream, want = random_stream(rng, n_haps=n_haps, n_ref_tx=n_ref, shape=shape, window=window)
            ctx.upload_proteome(proteome)
        rs = ctx.upload_stream(stream)
        for kernel in ((0, 6, 7) if fasta else (0, 6, 7, 8)):
            for slices in (0, 3):
                for variant in ((0, 24) if kernel in (0, 6) else (0,)):            # 24: the padded form whatever the rule says
                    ctx.set_launch_opts(variant=variant)
                    print('cfg', seed, shape, n_haps, n_ref, window, fasta, kernel, slices, variant, file=sys.stderr, flush=True)
                    b = ctx.batch()
                    try:
                        try:
                            b.build_and_execute(rs, kernel, slices)
                        except V2PError as e:
                            if not (kernel == 6 and e.code == -9):
                                raise
                            b.reset(); b.build_and_execute(rs, 7, slices)
                        b.sync()
                        if os.environ.get('FUZZ_TRACE'): print(' called', b.image_form(), file=sys.stderr, flush=True)
                        for rep in range(3):                                         # first execute, then twice the re-execution form
                            for h, w in enumerate(want):
                                got = b.download_hap(h)
                                if got.size != w.size or not np.array_equal(got, w):
                                    bad.append({"seed": seed, "fasta": fasta, "shape": shape, "kernel": kernel, "slices": slices, "variant": variant, "rep": rep, "hap": h}); break
                            if os.environ.get('FUZZ_TRACE'): print(' checked', rep, file=sys.stderr, flush=True)
                            b.execute(); b.sync()
                            if os.environ.get('FUZZ_TRACE'): print(' executed', rep, b.image_form(), file=sys.stderr, flush=True)
                        runs += 1
                    except Exception as e:                                           # noqa: BLE001
                        bad.append({"seed": seed, "fasta": fasta, "shape": shape, "kernel": kernel, "slices": slices, "variant": variant, "error": repr(e)[:300]})
                    finally:
                        b.close()
        ctx.set_launch_opts()
        rs.close()
print(json.dumps({"seeds": [first, first + count], "runs": runs, "failures": bad[:20], "n_failures": len(bad)}))
sys.exit(1 if bad else 0)
