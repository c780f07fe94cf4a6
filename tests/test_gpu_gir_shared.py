"""v2p_execute_gir_shared: the Engine::GPU arm of GIR::execute entered from many threads on ONE context, as the reference's Rayon
workers enter it (parts/exec.rs:36-39, personalized_genome.rs:64-65): concurrent calls are coalesced into one upload / launch /
download; every result the oracle's, every panic the reference's, whatever company a call had in its batch."""
import json
import subprocess
import threading

import numpy as np
import pytest

from gen_util import oracle_run, random_gir, random_tape

pytestmark = pytest.mark.gpu


def _jobs(coracle, seed, n):
    """n GIRs of every kind: canonical (coalesced), with gaps (cells that must keep the caller's fill), overlapping / descending
    (ordered path), a tape with a char above 0xFF (4-byte path), an empty Task vector, and three that the reference would panic on."""
    rng = np.random.default_rng(seed)
    jobs = []
    for j in range(n):
        kind = j % 16
        n_ref, n_alt = int(rng.integers(40, 60000)), int(rng.integers(4, 3000))
        ref, alt = random_tape(rng, n_ref), random_tape(rng, n_alt)
        g = random_gir(rng, int(rng.integers(1, 4000)), ref.size, alt.size, mean_len=int(rng.integers(2, 300)), p_gap=0.2 if kind in (3, 11) else 0.0)
        fill = ord("x") if kind == 3 else ord(".")
        expect_err = None
        if kind == 5:                                    # ordered: the last two tasks rewrite the head of the tape
            k = min(3, g["code"].size)
            g["start_pos_res"][-k:] = np.arange(k, dtype=np.uint64) * 0
            g["length"][-k:] = np.minimum(g["length"][-k:], 3)
        elif kind == 7:                                  # wide chars
            ref = ref.copy()
            ref[int(rng.integers(0, ref.size))] = 0x1F600
        elif kind == 9:
            g = dict(code=np.zeros(0, np.uint8), start_pos=np.zeros(0, np.uint64), length=np.zeros(0, np.uint64), start_pos_res=np.zeros(0, np.uint64), n_res=int(rng.integers(0, 50)))
        elif kind == 13:                                 # task.rs:43: source slice out of range -> the reference panics at this row
            row = int(rng.integers(0, g["code"].size))
            g["start_pos"][row] = (ref.size if g["code"][row] == 0 else alt.size) + 1
            g["length"][row] = max(int(g["length"][row]), 1)
            expect_err = ("V2P_ERR_SRC_OOB", row)
        want = None
        if expect_err is None:
            if kind == 5:
                t = coracle.pack_tasks(g["code"], g["start_pos"], g["length"], g["start_pos_res"])
                want = coracle.gir_execute(t, ref, alt, np.full(g["n_res"], fill, dtype=np.uint32))
            else:
                want = oracle_run(coracle, g, ref, alt, fill=fill)
        jobs.append(dict(g=g, ref=ref, alt=alt, fill=fill, want=want, err=expect_err, kind=kind))
    return jobs


@pytest.mark.parametrize("n_threads,n_jobs,seed,two_halves", [(64, 640, 1, False), (16, 200, 2, False), (3, 48, 3, False), (16, 320, 4, True), (64, 512, 5, True)])
def test_threads_on_one_context_are_coalesced_and_exact(gpu_ctx, coracle, n_threads, n_jobs, seed, two_halves):
    """two_halves: every worker keeps TWO calls in flight -- v2p_gir_submit for job k + 1 before v2p_gir_collect of job k
    (personalized_genome.rs:64-65: a sample's two haplotypes) -- every kind of GIR, every panic, reported by collect."""
    from vcf2prot_amd._native import V2PError, ERR_NAMES
    jobs = _jobs(coracle, seed, n_jobs)
    nb0, nc0 = gpu_ctx.coalesce_stats()
    results, failures = [None] * n_jobs, []
    nxt, lock = [0], threading.Lock()

    def finish(item):
        j, tk, res = item
        try:
            gpu_ctx.gir_collect(tk)
            results[j] = res
        except V2PError as e:
            results[j] = (ERR_NAMES.get(e.code, e.code), e.index, res)
        except Exception as e:                           # noqa: BLE001
            failures.append((j, repr(e)))

    def worker():
        pending = []
        while True:
            with lock:
                j = nxt[0]
                nxt[0] += 1
            if j >= n_jobs:
                for item in pending:
                    finish(item)
                return
            job = jobs[j]
            g = job["g"]
            res = np.full(g["n_res"], job["fill"], dtype=np.uint32)
            if two_halves:
                try:
                    tk = gpu_ctx.gir_submit(g["code"], g["start_pos"], g["length"], g["start_pos_res"], job["ref"], job["alt"], res)
                    while tk is None:                    # V2P_BUSY: every batch in flight -- collect what this worker holds, then again
                        if pending:
                            finish(pending.pop(0))
                        tk = gpu_ctx.gir_submit(g["code"], g["start_pos"], g["length"], g["start_pos_res"], job["ref"], job["alt"], res)
                except Exception as e:                   # noqa: BLE001
                    failures.append((j, repr(e)))
                    continue
                pending.append((j, tk, res))
                if len(pending) == 2:
                    finish(pending.pop(0))
                continue
            try:
                gpu_ctx.execute_gir_shared(g["code"], g["start_pos"], g["length"], g["start_pos_res"], job["ref"], job["alt"], res)
                results[j] = res
            except V2PError as e:
                results[j] = (ERR_NAMES.get(e.code, e.code), e.index, res)
            except Exception as e:                       # noqa: BLE001
                failures.append((j, repr(e)))

    th = [threading.Thread(target=worker) for _ in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not failures, failures[:3]
    for j, job in enumerate(jobs):
        if job["err"] is not None:
            name, row, res = results[j]
            assert (name, row) == job["err"], (j, results[j][:2])
            assert (res == job["fill"]).all(), j          # nothing was written
        else:
            assert isinstance(results[j], np.ndarray), (j, job["kind"], results[j][:2] if results[j] is not None else None)
            assert np.array_equal(results[j], job["want"]), (j, job["kind"])
    nb, nc = gpu_ctx.coalesce_stats()
    joined = nc - nc0
    assert joined >= n_jobs * 9 // 16                     # the canonical, byte-char GIRs went through batches
    if n_threads >= 16:
        assert nb - nb0 < joined                          # ... and shared them


def test_harness_workers_sharing_one_context(built, coracle):
    """`v2p_harness run <preset> <n> <threads> --shared`: the C++ mirror's GIR::execute_shared from a worker pool on ONE GpuContext."""
    from vcf2prot_amd import build
    from vcf2prot_amd.cohort import Cohort
    harness = build.build_harness()
    for preset, n, threads, flag in (("C1", 8, 4, "--shared"), ("C3", 12, 6, "--shared"), ("C3", 24, 6, "--async"), ("C2", 16, 16, "--async")):
        p = subprocess.run([harness, "run", preset, str(n), str(threads), flag], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        out = json.loads(p.stdout.strip().split("\n")[-1])
        assert "shared" in out["mode"] and (flag == "--shared" or "v2p_gir_submit" in out["mode"])
        c = Cohort.preset(preset)
        for h in range(n):
            hap = c.haplotype(h)
            t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = coracle.gir_execute(t, c.ref_tape_u32(h), hap.alt.astype(np.uint32), np.full(hap.n_res, ord("."), dtype=np.uint32))
            assert out["digests"][h] == coracle.digest_u32(want), (preset, h)
