"""600 random single-transcript cases answered by the reference binary (oracle/make_random_kats.py ->
tests/golden/kat_random.json; every other case with neighbouring / colliding mutation ranges): the Instruction list, the
Vec<Task> and the FASTA record it produced, or its abort -- 54 of the 55 aborts are slice panics inside Task::execute
(task.rs:44,48), i.e. on the hot path itself, with the Task vector already printed.

Steps 4a (restatement and C++), 4b (C++) and the executor (GPU) must reproduce all of it."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import frontend_oracle as F  # noqa: E402


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(HERE, "golden", "kat_random.json")) as f:
        return json.load(f)["cases"]


def chain(c):
    """('panic' | 'skip' | instruction dicts, 4b status, tasks, alt, res_len) of the product for one case (checks off, like the binary)."""
    from vcf2prot_amd import step4a
    from vcf2prot_amd.step4b import transcript_g_rep
    groups = F.group_muts_per_transcript(c["mutations"])
    assert len(groups) == 1
    muts = groups[0][1]
    try:
        o = F.transcript_instructions(c["transcript"], muts, inspect=False, panic_inspect=False)
        o = "skip" if o is None else [i.as_dict() for i in o]
    except F.ReferencePanic:
        o = "panic"
    rc, p = step4a.transcript_instructions([(m.mut_type, m.ref_aa_position, m.mut_aa_position, m.ref_aa, m.mut_aa) for m in muts], 0)
    p = {0: p, 1: "skip", 2: "panic"}[rc]
    assert o == p, c["name"]                                   # restatement == C++ on every case
    if not isinstance(p, list):
        return p, None, None, None, None
    rc, t, alt, res_len = transcript_g_rep(p, len(c["ref"]))
    return p, rc, t, alt, res_len


def test_steps_4a_4b_reproduce_the_reference(built, cases):
    n_ok = n_abort_early = n_abort_exec = 0
    codes = set()
    for c in cases:
        ins, rc, t, alt, res_len = chain(c)
        if not c["panics"]:
            assert ins == c["instructions"], c["name"]
            assert rc == 0 and t.tolist() == c["tasks"] and res_len == len(c["record"]), c["name"]
            if c["tasks"]:
                assert alt.decode() == c["alt"], c["name"]
            codes |= {i["code"] for i in ins}
            n_ok += 1
        elif isinstance(ins, list) and rc == 0:
            # the reference printed this very Task vector and then died executing it
            assert c["panic_in_executor"] and ins == c["instructions"] and t.tolist() == c["tasks"], c["name"]
            n_abort_exec += 1
        else:
            n_abort_early += 1                                  # refused before the executor (usize arithmetic the release binary let wrap)
    assert n_ok >= 500 and n_abort_exec >= 10 and n_abort_early >= 10
    assert {"M", "N", "I", "J", "D", "C", "F", "R", "G", "X", "L", "0", "U", "K", "A", "B", "P", "T", "2", "3"} <= codes


@pytest.mark.gpu
def test_executor_reproduces_records_and_slice_panics(built, gpu_ctx, cases):
    from vcf2prot_amd import _native as N
    n_rec = n_panic = 0
    for c in cases:
        ins, rc, t, alt, res_len = chain(c)
        if not isinstance(ins, list) or rc != 0:
            continue
        ref = np.frombuffer(c["ref"].encode(), dtype=np.uint8).astype(np.uint32)
        a32 = np.frombuffer(alt, dtype=np.uint8).astype(np.uint32)
        out = np.full(res_len, ord("."), dtype=np.uint32)
        args = (t[:, 0].astype(np.uint8), t[:, 1], t[:, 2], t[:, 3], ref, a32, out)
        if c["panics"]:
            with pytest.raises(N.V2PError) as e:               # task.rs:44,48: the slice panics of Task::execute
                gpu_ctx.execute_gir(*args)
            assert e.value.code in (N.V2P_ERR_RES_OOB, N.V2P_ERR_SRC_OOB), c["name"]
            n_panic += 1
        else:
            gpu_ctx.execute_gir(*args)
            assert "".join(map(chr, out)) == c["record"], c["name"]
            n_rec += 1
    assert n_rec >= 500 and n_panic >= 10


def test_inspect_txp_flags_the_task_vectors_the_executor_dies_on(built, cases):
    """transcript_instructions.rs:386-421 (INSPECT_TXP, default on in the current source): the validation that would stop
    the inconsistent Task vectors one step before the executor."""
    from vcf2prot_amd.step4b import inspect_transcript_tasks
    n_bad = 0
    for c in cases:
        ins, rc, t, alt, res_len = chain(c)
        if not isinstance(ins, list) or rc != 0:
            continue
        status, at = inspect_transcript_tasks(t, res_len)
        if c["panics"]:
            assert status in (1, 2), c["name"]
            n_bad += 1
        else:
            assert status == 0, c["name"]              # every Task vector the reference executed tiles its result exactly
    assert n_bad >= 10
    assert inspect_transcript_tasks(np.array([[0, 0, 4, 0], [1, 0, 1, 4]]), 5) == (0, -1)
    assert inspect_transcript_tasks(np.array([[0, 0, 4, 0], [1, 0, 1, 5]]), 6) == (1, 1)
    assert inspect_transcript_tasks(np.array([[0, 0, 4, 0], [1, 0, 1, 4]]), 6) == (2, -1)
    assert inspect_transcript_tasks(np.zeros((0, 4)), 0) == (0, -1)


@pytest.mark.gpu
def test_all_records_through_the_step5_builder_with_fasta_emit(built, gpu_ctx, cases):
    """The 545 answered cases as transcripts of two haplotypes: un-rebased Task vectors into v2p_batch_add_transcript
    (step 5), record headers resident behind the proteome, one launch; the arena must be the FASTA text."""
    good = [c for c in cases if not c["panics"]]
    refs, off, pos = [], {}, 0
    for c in good:
        off[c["name"]] = pos
        refs.append(c["ref"])
        pos += len(c["ref"])
    hdr, hoff, hpos = ["\n"], {}, 1
    for c in good:
        for h in (1, 2):
            t = f">{c['transcript']}_{h}\n"
            hoff[(c["name"], h)] = (hpos, len(t))
            hdr.append(t)
            hpos += len(t)
    gpu_ctx.upload_reference(np.frombuffer("".join(refs).encode(), dtype=np.uint8), np.frombuffer("".join(hdr).encode(), dtype=np.uint8))
    b = gpu_ctx.batch()
    want = []
    for h, subset in ((1, good[0::2]), (2, good[1::3])):
        b.begin_haplotype()
        text = []
        for c in subset:
            ins, rc, t, alt, res_len = chain(c)
            ho, hl = hoff[(c["name"], h)]
            b.add_transcript(t[:, 0].astype(np.uint8), t[:, 1], t[:, 2], t[:, 3], off[c["name"]], len(c["ref"]),
                             np.frombuffer(alt, dtype=np.uint8), res_len, ho, hl)
            text.append(f">{c['transcript']}_{h}\n{c['record']}\n")
        b.end_haplotype()
        want.append("".join(text))
    b.finalize()
    b.execute()
    b.sync()
    for h in range(2):
        assert b.download_hap(h).tobytes().decode() == want[h]
    b.close()
