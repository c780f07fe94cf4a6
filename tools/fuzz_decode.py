#!/usr/bin/env python3
"""A wider net than the fixed seeds of tests/test_gpu_decode.py: random VCF text (tests/frontend_util.py: masks of one and several words,
'.', extra FORMAT fields, every genotype spelling) of random shape through the GPU bitmask decode, every haplotype's id list against the
restatement (oracle/frontend_oracle.py).    python tools/fuzz_decode.py [first_seed] [n_seeds]"""
import os, sys, json, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from vcf2prot_amd import build
build.build_all()
from vcf2prot_amd.engine import Context
from vcf2prot_amd.frontend import VcfIndex, decode_bitmasks
from frontend_util import oracle_lists, random_vcf

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad, runs = [], 0
with Context(0) as ctx:
    for seed in range(first, first + count):
        r = random.Random(seed * 7919)
        shape = r.choice([(r.randrange(1, 900), r.randrange(1, 120)), (r.randrange(1, 60), r.randrange(100, 3000)), (r.randrange(1, 300), r.randrange(1, 400))])
        p_zero = r.choice([0.0, 0.1, 0.5, 0.9, 0.99])
        cfg = {"seed": seed, "records": shape[0], "samples": shape[1], "p_zero": p_zero, "fmt_extra": r.random() < 0.5, "max_csq": r.choice([3, 16, 40, 70])}
        print("cfg", cfg, file=sys.stderr, flush=True)
        text = random_vcf(seed, shape[0], shape[1], max_csq=cfg["max_csq"], p_zero=p_zero, fmt_extra=cfg["fmt_extra"], unique_positions=False)
        try:
            want = oracle_lists(text)[4]
            idx = VcfIndex(text.encode())
            got = decode_bitmasks(ctx, idx)
            ok = got.n_haplotypes == len(want) and all(got.of(h).tolist() == w for h, w in enumerate(want))
            if not ok:
                bad.append(cfg)
        except Exception as e:                      # noqa: BLE001
            bad.append({**cfg, "error": repr(e)[:300]})
        runs += 1
    # ---- aborts: one to three malformed columns at random places; the error must be the FIRST offending column's, located ----
    from vcf2prot_amd import _native as N
    BAD = [("0|1:-7", -20), ("0|1:3,-1", -20), ("0|1:-0", -21), ("0|1:1,,1", -21), ("0|1:1,a", -21), ("0|1:5,4294967296", -21), ("0|1:64", -22), ("0|1:1,16", -22)]
    n_abort = 0
    for seed in range(first, first + (count if os.environ.get("FUZZ_ABORTS", "1") == "1" else 0)):
        r = random.Random(seed * 104729)
        n_rec, n_smp = r.randrange(1, 400), r.randrange(1, 300)
        text = random_vcf(seed, n_rec, n_smp, max_csq=3, p_zero=r.choice([0.0, 0.5, 0.95]), fmt_extra=r.random() < 0.5, unique_positions=False)
        lines = text.split("\n")
        f0 = next(i for i, ln in enumerate(lines) if ln and not ln.startswith("#"))
        places = sorted({(r.randrange(n_rec), r.randrange(n_smp)) for _ in range(r.randrange(1, 4))})
        kinds = [r.choice(BAD) for _ in places]
        for (rec, smp), (txt, _) in zip(places, kinds):
            ln = lines[f0 + rec].split("\t"); ln[9 + smp] = txt; lines[f0 + rec] = "\t".join(ln)
        cfg = {"seed": seed, "records": n_rec, "samples": n_smp, "places": places, "kinds": kinds}
        print("abort cfg", cfg, file=sys.stderr, flush=True)
        try:
            decode_bitmasks(ctx, VcfIndex("\n".join(lines).encode()))
            bad.append({**cfg, "error": "no abort"})
        except N.V2PError as e:
            if e.code != kinds[0][1] or e.index != places[0][0] * n_smp + places[0][1]:
                bad.append({**cfg, "got": [e.code, e.index]})
        n_abort += 1
print(json.dumps({"seeds": [first, first + count], "runs": runs, "abort_runs": n_abort, "failures": bad[:20], "n_failures": len(bad)}))
sys.exit(1 if bad else 0)
