#!/usr/bin/env python3
"""A wider net than the fixed seeds of tests/test_gpu_oneshot.py: random irregular transcript streams (tests/stream_util.py) through the ONE
call under every kernel choice and slicing, the image's re-execution forms (dense from padded, pieces from dense, staged) included --
every haplotype's bytes against the numpy expectation.    python tools/fuzz_one_call.py [first_seed] [n_seeds] [--development]
Default: the PRODUCT library and what it takes (kernel 0 / 6 / 7 / 9, one slice, no switches).  --development: libv2p_bench.so -- the same
engine compiled with the A/B switches, slices, PATCH images (kernel 8) and the grid builders of rounds 2-3 (csrc/bench/v2p_bench.h)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vcf2prot_amd import build
DEV = "--development" in sys.argv
sys.argv = [x for x in sys.argv if x != "--development"]
build.build_hip(); build.build_cohort()
if DEV:
    build.build_bench()
from vcf2prot_amd.engine import Context
from vcf2prot_amd._native import V2PError
from stream_util import random_stream

first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad, runs = [], 0
UNSUPPORTED_BY_NUMBER = (6, 8, 9)                       # an image kind asked for by number may refuse the stream (-9); 0 and 7 take every stream these generators make
with Context(0, development=DEV) as ctx:
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
        for kernel in ((0, 6, 7, 9) + ((8,) if DEV and not fasta else ())):
            for slices in ((0, 3) if DEV else (0, 1)):
                for variant in ((0, 24, 25, 23, 28) if DEV and kernel in (0, 6) else (0,)):    # 24: the padded form whatever the rule says; 25: dense images staged too; 23: no staging; 28: never a tile image
                    # (small phases -- 8+ chunks -- on some runs: the phased launches, the read-ahead riding on them and the two staging buffers also on these small images)
                    if (seed + kernel + slices + variant) % 2:
                        ctx.set_launch_opts(variant=variant, phase_bytes=int([2048, 16384, 1 << 18][(seed + variant) % 3]), phase_min_chunks=8)
                    else:
                        ctx.set_launch_opts(variant=variant)
                    print('cfg', seed, shape, n_haps, n_ref, window, fasta, kernel, slices, variant, file=sys.stderr, flush=True)
                    b = ctx.batch()
                    try:
                        try:
                            b.build_and_execute(rs, kernel, slices)
                        except V2PError as e:
                            if not (kernel in UNSUPPORTED_BY_NUMBER and e.code == -9):
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
                            b.scribble(); b.execute(); b.sync()
                            if os.environ.get('FUZZ_TRACE'): print(' executed', rep, b.image_form(), file=sys.stderr, flush=True)
                        runs += 1
                    except Exception as e:                                           # noqa: BLE001
                        bad.append({"seed": seed, "fasta": fasta, "shape": shape, "kernel": kernel, "slices": slices, "variant": variant, "error": repr(e)[:300]})
                    finally:
                        b.close()
        ctx.set_launch_opts()
        rs.close()
    # ---- the two-call builders on a HOST stream (v2p_batch_build_on_device): the grid builders of round 3 (kernels 1..5, a window size) and
    # the rows builder (6, 7, 0); a refusal must be the documented one (-9: the stream does not fit the image kind that was asked for) ----
    n_two = 0
    for seed in range(first, first + (count if os.environ.get("FUZZ_TWO_CALL", "1") == "1" else 0)):
        rng = np.random.default_rng(seed + 31337)
        shape = ("snv", "mix", "long")[seed % 3]
        window = int(rng.choice([4096, 8192]))
        if seed % 2:                                                                   # (odd seeds: FASTA emit through the two-call builders too)
            proteome, headers, stream, want = random_stream(rng, n_haps=int(rng.integers(1, 120)), n_ref_tx=int(rng.integers(1, 30)), shape=shape, window=window, fasta=True)
            ctx.upload_reference(proteome, headers)
        else:
            proteome, stream, want = random_stream(rng, n_haps=int(rng.integers(1, 120)), n_ref_tx=int(rng.integers(1, 30)), shape=shape, window=window)
            ctx.upload_proteome(proteome)
        # (a kind's own window sizes: per-block kernels multiples of 4 KiB up to 60 KiB, the dense kernel 4 / 8 / 12 KiB, wave kernels multiples of 1 KiB up to 10)
        windows = {1: [4096, 8192, 16384, 28672], 2: [4096, 8192, 16384, 32768], 3: [4096, 8192, 12288], 4: [1024, 2048, 4096, 10240], 5: [2048, 3072, 4096, 6144, 10240]}
        for kernel in ((1, 2, 3, 4, 5, 6, 7) if DEV else (6, 7)):
            window = int(rng.choice(windows[kernel])) if kernel in windows else 0
            cfg = {"seed": seed, "shape": shape, "window": window, "kernel": kernel, "two_call": True}
            print("two cfg", cfg, file=sys.stderr, flush=True)
            b = ctx.batch()
            try:
                try:
                    b.build_on_device(stream, window, kernel)
                except V2PError as e:
                    if e.code != -9 or kernel == 7:
                        raise
                    continue                                                          # (an image kind asked for by number may not take the stream: too many descriptors in a window / a row)
                for rep in range(2):
                    b.scribble(); b.execute(); b.sync()
                    for h, w in enumerate(want):
                        got = b.download_hap(h)
                        if got.size != w.size or not np.array_equal(got, w):
                            bad.append({**cfg, "rep": rep, "hap": h}); break
                n_two += 1
            except Exception as e:                                                   # noqa: BLE001
                bad.append({**cfg, "error": repr(e)[:300]})
            finally:
                b.close()
    # ---- what the reference would panic on (update_task, haplotype_instruction.rs:154; Task::execute's slices, task.rs:43,47): one or two
    # violations injected into a clean stream -- the call must report the FIRST offending Task, with update_task's precedence, and the batch
    # must take a clean stream afterwards ----
    n_err = 0
    for seed in range(first, first + (count if os.environ.get("FUZZ_ERRORS", "1") == "1" else 0)):
        rng = np.random.default_rng(seed + 77777)
        shape = ("snv", "mix", "long")[seed % 3]
        proteome, stream, want = random_stream(rng, n_haps=int(rng.integers(1, 500)), n_ref_tx=int(rng.integers(1, 30)), shape=shape, window=4096)
        k = stream.keep
        n_tasks = stream.struct.n_tasks
        if n_tasks < 8:
            continue
        ctx.upload_proteome(proteome)
        good = ctx.upload_stream(stream)
        tb, rl, res, ab = k[4], k[2], k[3], k[5]
        code, sp, ln, sr = (a.copy() for a in k[6:10])
        picks = sorted(set(int(x) for x in rng.integers(0, n_tasks, size=int(rng.integers(1, 3)))))
        picks = [p_ for j, p_ in enumerate(picks) if j == 0 or p_ - picks[j - 1] > 2]
        expect = None
        for i in picks:
            t = int(np.searchsorted(tb, i, side="right") - 1)
            kind = str(rng.choice(["code", "res", "src", "order"]))
            if kind == "order" and (i == int(tb[t]) or int(sr[i - 1]) + int(ln[i - 1]) < 1 or int(code[i - 1]) > 1):
                kind = "code"
            if kind == "code":
                code[i] = int(rng.integers(2, 256)); err = -3
            elif kind == "res":
                sr[i] = int(res[t]) + 1; err = -4
            elif kind == "src":
                bound = int(rl[t]) if code[i] == 0 else int(ab[t + 1] - ab[t])
                sp[i] = bound + 1; err = -5
            else:
                sr[i] = int(sr[i - 1]) + int(ln[i - 1]) - 1; err = -6
            if expect is None:
                expect = (err, i, kind)
        from stream_util import Stream
        n_tx = stream.struct.n_tx
        badst = Stream(k[0], k[1], k[2], k[3], k[4], k[5], code[:n_tasks], sp[:n_tasks], ln[:n_tasks], sr[:n_tasks], k[10][:stream.struct.n_alt])
        rs = ctx.upload_stream(badst)
        for kernel in (0, 6, 7, 9):
            for slices in ((0, 3) if DEV else (0,)):
                cfg = {"seed": seed, "shape": shape, "kernel": kernel, "slices": slices, "expect": list(expect), "picks": picks}
                print("err cfg", cfg, file=sys.stderr, flush=True)
                b = ctx.batch()
                try:
                    try:
                        b.build_and_execute(rs, kernel, slices); b.sync()
                        bad.append({**cfg, "error": "no error reported"})
                    except V2PError as e:
                        if kernel in (6, 9) and e.code == -9:
                            pass                                                       # (a row with more than 64 descriptors, transcripts no tile holds: refused before the Tasks are judged)
                        elif e.code != expect[0] or e.index != expect[1]:
                            bad.append({**cfg, "got": [e.code, e.index]})
                    b.reset()
                    try:
                        b.build_and_execute(good, kernel, slices)
                    except V2PError as e:
                        if not (kernel in (6, 9) and e.code == -9):
                            raise
                        b.reset(); b.build_and_execute(good, 7, slices)
                    b.sync()
                    if not all(np.array_equal(b.download_hap(h), w) for h, w in enumerate(want)):
                        bad.append({**cfg, "error": "the clean stream behind the refused one gave other bytes"})
                    n_err += 1
                except Exception as e:                                               # noqa: BLE001
                    bad.append({**cfg, "error": repr(e)[:300]})
                finally:
                    b.close()
        rs.close(); good.close()
print(json.dumps({"seeds": [first, first + count], "runs": runs, "two_call_runs": n_two, "error_runs": n_err, "failures": bad[:20], "n_failures": len(bad)}))
sys.exit(1 if bad else 0)
