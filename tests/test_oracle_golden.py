"""CPU suite: the oracle against the reference's known answers and harvested vectors.

These tests pin oracle/sir_oracle.{py,c}; they never touch the HIP engine.
Reference line numbers are under /root/reference/src/data_structures/InternalRep.
"""
import numpy as np
import pytest

from sir_oracle import (COracle, OraclePanic, Task, TranscriptGIR, fasta_records, gir_execute, haplotype_concat,
                        str_to_u32, task_execute, u32_to_str, validate_contiguity)


def test_task_rs_test_execute_python():
    # task.rs:118-144
    ref = list("ABCFEFGH")
    alt = list(reversed(ref))
    res = ["x"] * 10
    exp = ["x"] * 10
    task_execute(Task(0, 1, 1, 8), res, ref, alt)
    exp[8] = "B"
    assert res == exp
    task_execute(Task(0, 4, 1, 4), res, ref, alt)
    exp[4] = "E"
    assert res == exp
    task_execute(Task(0, 6, 2, 6), res, ref, alt)
    exp[6], exp[7] = "G", "H"
    assert res == exp


def test_task_rs_test_execute_c(coracle):
    ref = str_to_u32("ABCFEFGH")
    alt = ref[::-1].copy()
    res = np.full(10, ord("x"), dtype=np.uint32)
    t = coracle.pack_tasks([0, 0, 0], [1, 4, 6], [1, 1, 2], [8, 4, 6])
    coracle.gir_execute(t, ref, alt, res)
    assert u32_to_str(res) == "xxxxExGHBx"


def test_gir_rs_doc_example(coracle):
    # gir.rs:172-196: ref TEST, alt G, tasks (0,0,4,0),(1,0,1,4) -> TESTG
    res = gir_execute([Task(0, 0, 4, 0), Task(1, 0, 1, 4)], list("TEST"), list("G"), ["."] * 5)
    assert "".join(res) == "TESTG"
    t = coracle.pack_tasks([0, 1], [0, 0], [4, 1], [0, 4])
    out = coracle.gir_execute(t, str_to_u32("TEST"), str_to_u32("G"), np.full(5, ord("."), dtype=np.uint32))
    assert u32_to_str(out) == "TESTG"


def test_alt_code_is_any_nonzero():
    # task.rs:42-49: only exe_code == 0 selects the reference tape
    res = ["."] * 2
    task_execute(Task(7, 0, 2, 0), res, list("AB"), list("CD"))
    assert res == ["C", "D"]


def test_engine_gpu_arm_panics_in_reference():
    with pytest.raises(OraclePanic):
        gir_execute([], [], [], [], engine="gpu")          # gir.rs:236-239
    with pytest.raises(ValueError):
        gir_execute([], [], [], [], engine="cuda")         # engines.rs:27


def test_bounds_panics(coracle):
    with pytest.raises(OraclePanic):
        task_execute(Task(0, 0, 5, 0), ["."] * 4, list("ABCDE"), [])
    with pytest.raises(OraclePanic):
        task_execute(Task(0, 3, 5, 0), ["."] * 8, list("ABCDE"), [])
    t = coracle.pack_tasks([0, 0], [0, 3], [2, 5], [0, 2])
    with pytest.raises(OraclePanic) as e:
        coracle.gir_execute(t, str_to_u32("ABCDE"), str_to_u32(""), np.full(8, 46, dtype=np.uint32))
    assert e.value.index == 1 and "src_oob" in str(e.value)
    t = coracle.pack_tasks([0], [0], [5], [4])
    with pytest.raises(OraclePanic) as e:
        coracle.gir_execute(t, str_to_u32("ABCDE"), str_to_u32(""), np.full(8, 46, dtype=np.uint32))
    assert "res_oob" in str(e.value)


def test_debug_cpu_exec_predicate(coracle):
    # gir.rs:208-226
    ok = [Task(0, 0, 4, 0), Task(1, 0, 0, 4), Task(0, 4, 3, 4)]
    assert validate_contiguity(ok) == -1
    bad = [Task(0, 0, 4, 0), Task(1, 0, 1, 5)]
    assert validate_contiguity(bad) == 1
    with pytest.raises(OraclePanic) as e:
        gir_execute(bad, list("ABCDEFGH"), list("X"), ["."] * 8, debug_cpu_exec=True)
    assert e.value.index == 1
    t = coracle.pack_tasks([0, 1], [0, 0], [4, 1], [0, 5])
    assert coracle.validate(t) == 1
    with pytest.raises(OraclePanic):
        coracle.gir_execute(t, str_to_u32("ABCDEFGH"), str_to_u32("X"), np.full(8, 46, dtype=np.uint32), debug_cpu_exec=True)
    # without the flag the same vector executes and leaves the uncovered cell as it was
    out = coracle.gir_execute(t, str_to_u32("ABCDEFGH"), str_to_u32("X"), np.full(8, 46, dtype=np.uint32))
    assert u32_to_str(out) == "ABCD.X.."


def test_golden_cases_both_restatements(golden, coracle):
    """Task vectors + FASTA harvested from the reference binary (oracle/make_golden.py)."""
    assert len(golden["cases"]) >= 36
    for c in golden["cases"]:
        tasks = [Task(*t) for t in c["tasks"]]
        res = gir_execute(tasks, list(c["ref"]), list(c["alt"]), ["."] * c["res_len"], debug_cpu_exec=True)
        assert "".join(res) == c["expected"], c["name"]
        t = coracle.pack_tasks(*(zip(*c["tasks"]) if c["tasks"] else ([], [], [], [])))
        out = coracle.gir_execute(t, str_to_u32(c["ref"]), str_to_u32(c["alt"]),
                                  np.full(c["res_len"], ord("."), dtype=np.uint32), debug_cpu_exec=True)
        assert u32_to_str(out) == c["expected"], c["name"]
        assert c["matches_source_unit_test"], c["name"]


def test_golden_unit_test_assertions(golden):
    """The assertions of transcript_instructions.rs:884-1594, re-evaluated on the oracle's output."""
    n_checked = 0
    for c in golden["cases"]:
        a = c["asserts"]
        if not a:
            continue
        tasks = [Task(*t) for t in c["tasks"]]
        seq = "".join(gir_execute(tasks, list(c["ref"]), list(c["alt"]), ["."] * c["res_len"]))
        if "len" in a:
            assert len(seq) == a["len"], c["name"]
        if "len_delta" in a:
            assert len(seq) == len(c["ref"]) + a["len_delta"], c["name"]
        for idx, ch in a.get("residues", {}).items():
            assert seq[int(idx)] == ch, c["name"]
        if "equal_except" in a:
            assert len(seq) == len(c["ref"])
            for p, (x, y) in enumerate(zip(seq, c["ref"])):
                if p not in a["equal_except"]:
                    assert x == y, c["name"]
        n_checked += 1
    assert n_checked >= 28


def test_gap_cell_keeps_dot(golden):
    # test_correct_translation_20: the 'P' instruction covers 37 of 38 cells; the last stays '.'
    c = next(x for x in golden["cases"] if x["name"] == "test_correct_translation_20")
    assert c["expected"].endswith(".") and sum(t[2] for t in c["tasks"]) == c["res_len"] - 1


def test_step5_concat_and_fasta(golden):
    """haplotype_instruction.rs:75-158 + personalized_genome.rs:90-113 on the harvested transcripts."""
    girs, expected = [], []
    for i, c in enumerate(golden["cases"]):
        name = f"TX{i:03d}"
        girs.append(TranscriptGIR(name, [Task(*t) for t in c["tasks"]], c["alt"], c["ref"] if c["tasks"] else "", c["res_len"]))
        expected.append((f"{name}_1", c["expected"]))
    tasks, ann, alt, ref, res = haplotype_concat(girs)
    assert validate_contiguity([t for t in tasks]) in (-1, validate_contiguity(tasks))
    gir_execute(tasks, ref, alt, res)
    assert fasta_records(res, ann, 1) == sorted(expected)
    # start-lost transcript: annotation (k,k) and an empty record (transcript_instructions.rs:338-343)
    i0 = next(i for i, c in enumerate(golden["cases"]) if c["name"] == "appendix_start_lost")
    a, b = ann[f"TX{i0:03d}"]
    assert a == b


def test_digest_matches_definition(coracle):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, size=1000, dtype=np.uint8)

    def mix(x):
        M = (1 << 64) - 1
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    # include/vcf2prot_hip.h (v2p_batch_digests): sum_i (byte_i + 1) * 2^(8 * (i mod 8)) * splitmix64(i div 8)  mod 2^64
    for n in (1000, 1003, 8, 7, 1, 0):
        want = sum(((int(v) + 1) << (8 * (i & 7))) * mix(i >> 3) for i, v in enumerate(a[:n])) & ((1 << 64) - 1)
        assert coracle.digest_u8(a[:n]) == want, n
        assert coracle.digest_u32(a[:n].astype(np.uint32)) == want, n


def test_mt_driver_equals_sequential(coracle):
    """parts/exec.rs:34-40 equivalent: thread pool over haplotypes == one by one."""
    from gen_util import random_gir, random_tape, oracle_run
    rng = np.random.default_rng(11)
    jobs, want = [], []
    for h in range(12):
        ref, alt = random_tape(rng, 5000), random_tape(rng, 300)
        g = random_gir(rng, 400, ref.size, alt.size, p_gap=0.05)
        want.append(oracle_run(coracle, g, ref, alt).copy())
        t = coracle.pack_tasks(g["code"], g["start_pos"], g["length"], g["start_pos_res"])
        jobs.append((t, ref, alt, np.zeros(g["n_res"], dtype=np.uint32)))
    secs = coracle.mt_execute(jobs, n_threads=4, wide=True, reps=2)
    assert secs > 0
    for (t, ref, alt, res), w in zip(jobs, want):
        assert np.array_equal(res, w)
