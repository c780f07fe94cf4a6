"""GPU suite: the C++ host mirror (vcf2prot_amd/csrc/host/ppgg_gpu.hpp) and harness (the role of
the reference's exec::execute, parts/exec.rs:23-42) on top of the C ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(built):
    from vcf2prot_amd import build
    return build.build_harness()


def test_cpp_mirror_known_answers(harness):
    # task.rs:118-144, gir.rs:172-196, bounds panic, Engine::from_str -- through GIR::execute(Engine::GPU) in C++
    p = subprocess.run([harness, "kat"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "kat: ok" in p.stdout, p.stdout + p.stderr


@pytest.mark.parametrize("preset,n,threads", [("C1", 8, 3), ("C3", 6, 4)])
def test_harness_thread_pool_matches_oracle(harness, coracle, preset, n, threads):
    """Haplotype GIRs executed concurrently from several host threads (one ctx each), as Rayon
    workers enter GIR::execute; every result digest must equal the oracle's."""
    from vcf2prot_amd.cohort import Cohort
    p = subprocess.run([harness, "run", preset, str(n), str(threads)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = json.loads(p.stdout.strip().split("\n")[-1])
    c = Cohort.preset(preset)
    assert out["haplotypes"] == n
    for h in range(n):
        hap = c.haplotype(h)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        want = coracle.gir_execute(t, c.ref_tape_u32(h), hap.alt.astype(np.uint32), np.full(hap.n_res, ord("."), dtype=np.uint32))
        assert out["digests"][h] == coracle.digest_u32(want), (preset, h)
