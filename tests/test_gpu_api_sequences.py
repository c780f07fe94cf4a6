"""GPU suite: random call sequences against one v2p_batch.  Whatever the order, a call either succeeds or returns a
status (V2PError) -- never a crash -- and every batch that gets through finalize/execute holds exactly the oracle's
bytes for the haplotypes that were added."""
import random

import numpy as np
import pytest

from gen_util import oracle_run, random_gir, random_tape

pytestmark = pytest.mark.gpu


def test_random_batch_call_sequences(built, gpu_ctx, coracle):
    from vcf2prot_amd import _native as N
    rng = random.Random(5)
    nrng = np.random.default_rng(5)
    ref, alt = random_tape(nrng, 6000), random_tape(nrng, 600)
    girs = [random_gir(nrng, int(nrng.integers(1, 300)), ref.size, alt.size, mean_len=int(nrng.integers(2, 80)), p_gap=0.05) for _ in range(6)]
    wants = [oracle_run(coracle, g, ref, alt).astype(np.uint8) for g in girs]
    n_errors = n_checked = 0
    for trial in range(60):
        b = gpu_ctx.batch()
        added, finalized, executed = [], False, False
        for step in range(rng.randint(1, 9)):
            op = rng.choice(["add", "add", "add", "finalize", "execute", "sync", "download", "counts", "digests", "bad_add", "hap_range"])
            try:
                if op == "add":
                    k = rng.randrange(len(girs))
                    g = girs[k]
                    b.add_gir(g["code"], g["start_pos"], g["length"], g["start_pos_res"], ref, alt, g["n_res"])
                    assert not finalized, "add after finalize must be refused"
                    added.append(k)
                elif op == "bad_add":
                    g = girs[0]
                    sp = g["start_pos"].copy()
                    sp[0] = ref.size + alt.size + 5                      # reads past its tape: task.rs:43,47 would panic
                    b.add_gir(g["code"], sp, g["length"], g["start_pos_res"], ref, alt, g["n_res"])
                    assert g["length"][0] == 0, "an out-of-bounds task must be refused"
                    added.append(0)
                elif op == "finalize":
                    b.finalize()
                    finalized = True
                elif op == "execute":
                    b.execute()
                    assert finalized, "execute before finalize must be refused"
                    executed = True
                elif op == "sync":
                    b.sync()
                elif op == "counts":
                    assert b.counts()["n_haps"] == len(added) or not finalized
                elif op == "hap_range":
                    b.hap_range(rng.randrange(0, 4))
                elif op == "digests":
                    b.digests()
                elif op == "download":
                    h = rng.randrange(0, max(1, len(added)))
                    out = b.download_hap(h)
                    if executed and h < len(added):
                        b.sync()
                        assert np.array_equal(b.download_hap(h), wants[added[h]]), (trial, step)
                        n_checked += 1
                    del out
            except N.V2PError:
                n_errors += 1
        if not finalized:
            b.finalize()
            finalized = True
        if not executed:
            b.execute()
            executed = True
        if executed:
            b.sync()
            for h, k in enumerate(added):
                assert np.array_equal(b.download_hap(h), wants[k]), (trial, h)
                n_checked += 1
        b.close()
    assert n_errors > 20 and n_checked > 40
