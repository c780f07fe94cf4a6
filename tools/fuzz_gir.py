#!/usr/bin/env python3
"""The GIR-faithful arm under random Task vectors (tests/gen_util.py: canonical, gapped, zero-length, long tasks): v2p_execute_gir, and
v2p_execute_gir_shared / v2p_gir_submit + v2p_gir_collect from 16 threads on ONE context (calls coalesced into shared batches), every
result tape against the C oracle.    python tools/fuzz_gir.py [first_seed] [n_seeds]"""
import os, sys, json
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from vcf2prot_amd import build
build.build_all()
import sir_oracle
sir_oracle.build_c_oracle()
from sir_oracle import COracle
from vcf2prot_amd.engine import Context
from vcf2prot_amd._native import V2PError
from gen_util import random_gir, random_tape, oracle_run

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 40
orc = COracle()
bad, runs = [], 0


def case(seed):
    rng = np.random.default_rng(seed)
    ref, alt = random_tape(rng, int(rng.integers(1, 60000))), random_tape(rng, int(rng.integers(1, 5000)))
    n_tasks = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 256, 257, 1000, 5000, 20000]))
    g = random_gir(rng, n_tasks, ref.size, alt.size, mean_len=float(rng.choice([2, 6, 40, 200, 2000])), p_zero=float(rng.choice([0.0, 0.1, 0.5])),
                   p_gap=float(rng.choice([0.0, 0.0, 0.2])), p_alt=float(rng.choice([0.0, 0.4, 1.0])))
    return g, ref, alt, oracle_run(orc, g, ref, alt)


with Context(0) as ctx:
    for seed in range(first, first + count):
        g, ref, alt, want = case(seed)
        print("cfg", seed, g["code"].size, g["n_res"], file=sys.stderr, flush=True)
        res = np.full(g["n_res"], ord("."), dtype=np.uint32)
        try:
            ctx.execute_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, res)
            if not np.array_equal(res, want):
                bad.append({"seed": seed, "mode": "execute_gir"})
        except V2PError as e:
            bad.append({"seed": seed, "mode": "execute_gir", "error": repr(e)[:200]})
        runs += 1
    # many threads, one context: shared (blocking) and submit / collect with two in flight per worker
    cases = {s: case(s) for s in range(first, first + count)}

    def shared(s):
        g, ref, alt, want = cases[s]
        res = np.full(g["n_res"], ord("."), dtype=np.uint32)
        ctx.execute_gir_shared(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, res)
        return s, bool(np.array_equal(res, want))

    def pipelined(w):
        out, inflight = [], []
        for s in list(cases)[w::16]:
            g, ref, alt, want = cases[s]
            res = np.full(g["n_res"], ord("."), dtype=np.uint32)
            while True:
                tk = ctx.gir_submit(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, res)
                if tk is not None:
                    inflight.append((s, tk, want))
                    break
                if inflight:                                # V2P_BUSY: every batch of the queue is in flight -- collect one of ours first
                    s0, t0, w0 = inflight.pop(0); out.append((s0, bool(np.array_equal(ctx.gir_collect(t0), w0))))
            if len(inflight) > 1:
                s0, t0, w0 = inflight.pop(0); out.append((s0, bool(np.array_equal(ctx.gir_collect(t0), w0))))
        for s0, t0, w0 in inflight:
            out.append((s0, bool(np.array_equal(ctx.gir_collect(t0), w0))))
        return out

    with ThreadPoolExecutor(16) as pool:
        for rep in range(3):
            for s, ok in pool.map(shared, list(cases)):
                runs += 1
                if not ok:
                    bad.append({"seed": s, "mode": "shared", "rep": rep})
        try:
            for part in pool.map(pipelined, range(16)):
                for s, ok in part:
                    runs += 1
                    if not ok:
                        bad.append({"seed": s, "mode": "submit/collect"})
        except V2PError as e:
            bad.append({"mode": "submit/collect", "error": repr(e)[:300]})
print(json.dumps({"seeds": [first, first + count], "runs": runs, "failures": bad[:20], "n_failures": len(bad)}))
sys.exit(1 if bad else 0)
